#!/usr/bin/env python3
"""The LDS-staged few-row GEMM (gemm_f32_small_lds_kernel) against the register-fragment kernel it replaces: run with
VITA_SMALL_NO_LDS=1 and =0, `--save f` writes the outputs of a shape set, `--compare f` checks them bitwise; both print
HIP-event times per shape and the B = 128 training step."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(128, 2048, 512), (128, 512, 2048), (128, 512, 512), (25, 512, 512), (100, 336, 512), (128, 512, 256), (77, 2048, 512),
          (128, 512, 1024), (512, 512, 512), (33, 96, 132)]
outs, times = {}, {}
for (M, N, K) in shapes:
    x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev); r = torch.randn(M, N, generator=g).to(dev); gate = torch.randn(N, generator=g).to(dev)
    wc = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev); z = torch.randn(M, N, generator=g).to(dev)
    f1 = lambda: ops.linear(x, w, b, L.ACT_GELU_TANH)
    f2 = lambda: ops.linear(x, w, b, L.ACT_NONE, gate=gate, resid=r)
    f3 = lambda: ops.linear_t(x, wc, b_col=True, M=M, N=N, K=K, act=L.ACT_GELU_TANH, act_grad_z=z)
    for name, f in (("fwd_gelu", f1), ("fwd_gate_resid", f2), ("dgrad_bcol", f3)):
        y = f()
        torch.cuda.synchronize()
        outs[f"{M}x{N}x{K}/{name}"] = y.cpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            f()
        e0.record()
        for _ in range(50):
            f()
        e1.record(); e1.synchronize()
        times[f"{M}x{N}x{K}/{name}"] = round(e0.elapsed_time(e1) / 50 * 1e3, 2)
print(json.dumps({"no_lds": os.environ.get("VITA_SMALL_NO_LDS", "0"), "us_per_call": times}))
if "--save" in sys.argv:
    torch.save(outs, sys.argv[sys.argv.index("--save") + 1])
if "--compare" in sys.argv:
    ref = torch.load(sys.argv[sys.argv.index("--compare") + 1], weights_only=True)
    bad = [k for k in outs if not torch.equal(outs[k], ref[k])]
    print(json.dumps({"bitwise_equal": not bad, "differ": bad}))
    if bad:
        sys.exit(1)
