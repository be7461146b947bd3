"""Times vita_linear_bf16 (row operands) on the bf16 training step's 4096-row shapes.  VITA_BF16_CFG selects lab tile variants."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import _lib, ops
_lib.load()
dev = "cuda:0"
for M, N, K in ((4096, 512, 2048), (4096, 512, 512), (4096, 2048, 512), (8192, 512, 512)):
    x = ops.pack_bf16(torch.randn(M, K, device=dev))
    w = ops.pack_bf16(torch.randn(N, K, device=dev) * 0.05)
    b = torch.randn(N, device=dev)
    t_end = time.time() + 1.0
    while time.time() < t_end:
        ops.linear_bf16(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.linear_bf16(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(json.dumps({"cfg": os.environ.get("VITA_BF16_CFG", "0"), "M": M, "N": N, "K": K, "us": round(ms * 1e3, 2), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
