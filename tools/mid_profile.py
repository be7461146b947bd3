#!/usr/bin/env python3
"""generate_actions at one mid batch size, for rocprofv3 --kernel-trace: usage python3 tools/mid_profile.py [B] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
obs = torch.randn(B, 533).to(dev)
out = torch.empty(B, 336, device=dev)
for _ in range(20):
    ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
torch.cuda.synchronize()
print(f"B={B} generate_actions_ms={(time.perf_counter() - t0) / n * 1e3:.4f}")
