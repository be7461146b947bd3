#!/usr/bin/env python3
"""Serving-size latency of generate_actions / ode_sample: the weights-resident launch vs the per-launch path."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)

def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for B in (1, 5, 25, 32, 64):
    obs = torch.randn(B, 533).to(dev)
    x0 = torch.randn(B, 512).to(dev)
    out = torch.empty(B, 336, device=dev)
    row = {"B": B, "K": 6}
    for name, on in (("resident", True), ("per_launch", False)):
        ops.set_persistent_sampler(on)
        row[f"generate_actions_ms_{name}"] = round(timed(lambda: ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)), 4)
        row[f"ode_sample_ms_{name}"] = round(timed(lambda: ops.ode_sample(fw, x0, 6)), 4)
    ops.set_persistent_sampler(True)
    row["status"] = ops.ode_sample_status(fw, B, 6, dev)
    print(json.dumps(row), flush=True)
