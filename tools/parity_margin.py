#!/usr/bin/env python3
"""Normalised max error of generate_actions vs the fp64 oracle (row sample), K = 6 and 100: how much of the 1e-5 budget is used."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vita_oracle as O
from vita_amd import ops
dev = torch.device("cuda:0")
for task, Od, A in (("hook_package", 533, 21), ("robomimic_square", 555, 7)):
    sd = O.make_state_dict(Od, A, seed=0)
    sd64 = {k: v.double() for k, v in sd.items()}
    fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(4096, Od, generator=g)
    idx = torch.arange(0, 4096, 128)
    for K in (6, 100):
        ref = O.generate_actions(sd64, obs[idx].double(), K, 16, A).reshape(len(idx), -1)
        got = ops.generate_actions(sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev), fw, dw, obs.to(dev), K)[idx.to(dev)].cpu().double()
        e = ((got - ref).abs().max() / ref.abs().max()).item()
        ref32 = O.generate_actions(sd, obs[idx], K, 16, A).reshape(len(idx), -1).double()
        e32 = ((ref32 - ref).abs().max() / ref.abs().max()).item()
        print(f"{task} K={K}: hip vs fp64 {e:.3g}   (torch-cpu fp32 vs fp64 {e32:.3g})")
