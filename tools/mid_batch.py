#!/usr/bin/env python3
"""generate_actions latency at the batch sizes between the weights-resident launch (<= 64 rows) and the persistent launch
(multiples of 256 from 1024): the per-launch path on the few-row kernels."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)

def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for B in (65, 100, 128, 256, 512, 768, 1000):
    obs = torch.randn(B, 533).to(dev)
    out = torch.empty(B, 336, device=dev)
    ms = timed(lambda: ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out))
    print(json.dumps({"B": B, "K": 6, "generate_actions_ms": round(ms, 4), "chunks_per_s": round(B / ms * 1e3)}), flush=True)
