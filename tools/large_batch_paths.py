#!/usr/bin/env python3
"""generate_actions at multiples of 256 rows: the persistent launch vs the per-launch path."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)

def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for B in (1024, 1280, 1536, 2048, 3072, 4096, 8192):
    obs = torch.randn(B, 533).to(dev)
    out = torch.empty(B, 336, device=dev)
    row = {"B": B}
    for name, on in (("persistent", True), ("per_launch", False)):
        ops.set_persistent_sampler(on)
        ms = timed(lambda: ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out))
        row[name + "_ms"] = round(ms, 4)
    ops.set_persistent_sampler(True)
    row["takes_persistent"] = int(ops.ode_sample_takes_persistent(fw, B)) if hasattr(ops, "ode_sample_takes_persistent") else None
    print(json.dumps(row), flush=True)
