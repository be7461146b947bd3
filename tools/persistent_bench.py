"""Latent-only sampler (B,D)->(B,D): the persistent launch against the per-launch path, HIP events on the launch stream."""
import json
import sys

import torch

sys.path.insert(0, ".")
from vita_amd import ops  # noqa: E402
from vita_amd.synthetic import make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
sd = make_state_dict(533, 21, hidden_dim=H, seed=0)
fw = ops.FlowNetHandle(sd, dev)
x0 = torch.randn(B, 512, generator=torch.Generator().manual_seed(1)).to(dev)
f_step = 2 * fw.D * fw.H + fw.L * 4 * fw.H * fw.FF + 2 * fw.H * fw.D
out = {}
for name, on in (("per_launch", False), ("persistent", True), ("per_launch_again", False), ("persistent_again", True)):
    ops.set_persistent_sampler(on)
    for _ in range(3):
        ops.ode_sample(fw, x0, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.ode_sample(fw, x0, K)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = B * K * f_step / (ms * 1e-3) / 1e12
    out[name] = {"ms_per_solve": round(ms, 4), "tflops": round(tf, 2), "frac_of_f32_mfma_peak": round(tf / 157.3, 4)}
out["status"] = ops.ode_sample_status(fw, B, K, dev)
print(json.dumps({"B": B, "K": K, "H": H, **out}))
