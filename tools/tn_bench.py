"""Times vita_linear_bf16_tn (dW = dY^T X on row-major bf16 operands) on the training step's weight-gradient shapes.
usage: python tools/tn_bench.py [R]"""
import json
import sys
import time

import torch

from vita_amd import _lib, ops

_lib.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 28672
dev = "cuda:0"
for M, N in ((2048, 512), (512, 2048), (512, 512), (336, 512)):
    a = ops.pack_bf16(torch.randn(R, M, device=dev))
    b = ops.pack_bf16(torch.randn(R, N, device=dev))
    t_end = time.time() + 1.5
    while time.time() < t_end:
        ops.linear_bf16_tn(a, b, M, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear_bf16_tn(a, b, M, N)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"R": R, "M": M, "N": N, "ms": round(ms, 4), "tflops": round(2.0 * R * M * N / ms / 1e9, 1)}), flush=True)
