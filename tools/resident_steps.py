#!/usr/bin/env python3
"""Per-step and fixed cost of the weights-resident sampler: ode_sample at K = 1, 2, 6, 12, 24."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw = ops.FlowNetHandle(sd, dev)

def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for B in (1, 8, 16, 25, 32):
    x0 = torch.randn(B, 512).to(dev)
    row = {"B": B}
    for K in (1, 2, 6, 12, 24):
        row[f"K{K}"] = round(timed(lambda: ops.ode_sample(fw, x0, K)), 4)
    row["us_per_step"] = round((row["K24"] - row["K12"]) / 12 * 1e3, 2)
    row["fixed_us"] = round((row["K12"] - 12 * (row["K24"] - row["K12"]) / 12) * 1e3, 1)
    print(json.dumps(row), flush=True)
