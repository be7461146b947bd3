#!/usr/bin/env python3
"""generate_actions at 4096 rows: persistent launch vs launch by launch, alternating, 5 rounds of 40 calls each."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict
dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
B = 4096
obs = torch.randn(B, 533).to(dev); out = torch.empty(B, 336, device=dev); x0 = torch.randn(B, 512).to(dev)
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {"persistent": [], "per_launch": [], "persistent_latent": [], "per_launch_latent": []}
for r in range(5):
    for name, on in (("persistent", True), ("per_launch", False)):
        ops.set_persistent_sampler(on)
        res[name].append(round(timed(lambda: ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)), 4))
        res[name + "_latent"].append(round(timed(lambda: ops.ode_sample(fw, x0, 6)), 4))
ops.set_persistent_sampler(True)
print(json.dumps(res))
