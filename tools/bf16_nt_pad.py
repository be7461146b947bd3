"""vita_linear_bf16 with padded row strides (ldx = K + pad): does a power-of-two row stride cost operand delivery?"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import _lib as L, ops
if os.environ.get("VITA_LAB_LIB"):
    L.LIB_PATH = os.environ["VITA_LAB_LIB"]
lib = L.load()
dev = "cuda:0"
for M, N, K in ((4096, 512, 2048), (4096, 512, 512)):
    for pad in (0, 64, 128, 72):
        ld = K + pad
        xw = torch.zeros(M, ld, dtype=torch.int16, device=dev); xw[:, :K] = ops.pack_bf16(torch.randn(M, K, device=dev))
        ww = torch.zeros(N, ld, dtype=torch.int16, device=dev); ww[:, :K] = ops.pack_bf16(torch.randn(N, K, device=dev) * 0.05)
        y = torch.empty(M, N, device=dev)
        a = L.LinearArgs()
        a.X, a.ldx, a.W, a.ldw = xw.data_ptr(), ld, ww.data_ptr(), ld
        a.Y, a.ldy = y.data_ptr(), N
        a.M, a.N, a.K, a.act = M, N, K, 0
        st = torch.cuda.current_stream().cuda_stream
        def run():
            L.check(lib.vita_linear_bf16(C.byref(a), None, 0, st), "x")
        t_end = time.time() + 0.7
        while time.time() < t_end:
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(json.dumps({"M": M, "N": N, "K": K, "pad": pad, "us": round(ms * 1e3, 2), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
