import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from vita_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(4096, 512, device=dev); m = torch.randn(512, device=dev); a = torch.randn(512, device=dev)
for _ in range(5): ops.layernorm_modulate(x, m, a, 1e-6)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.layernorm_modulate(x, m, a, 1e-6)
e1.record(); e1.synchronize()
print(sys.argv[1:] , "ln_mod 4096x512: %.2f us" % (e0.elapsed_time(e1) * 1e3 / 200))
