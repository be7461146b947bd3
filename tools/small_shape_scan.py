"""Few-row GEMM kernel time against its shape (rocprofv3 --kernel-trace --stats over this script gives the kernel times):
is a B = 128 GEMM bound by its operand bytes per CU, or by fixed latency?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import _lib, ops
_lib.load()
dev = "cuda:0"
M = 128
for N, K in ((2048, 512), (2048, 256), (2048, 128), (1024, 512), (512, 512), (8192, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(300):
        ops.linear(x, w, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        ops.linear(x, w, b, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) / 300 * 1e3:.2f} us per call (host-inclusive)", flush=True)
