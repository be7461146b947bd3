#!/usr/bin/env python3
"""Serving-size latency: generate_actions at B = eval_n_envs (25, default_policy.yaml:48) — eager launches vs one HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vita_oracle as O
from vita_amd import ops

dev = torch.device("cuda:0")
sd = O.make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
for B in (1, 25, 128, 512):
    obs = torch.randn(B, 533).to(dev)
    out = torch.empty(B, 336, device=dev)
    for _ in range(5):
        ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)   # warm the side stream's workspace
    torch.cuda.current_stream().wait_stream(s)
    ref = out.clone()
    with torch.cuda.graph(g, stream=s):
        ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50 * 1e3
    print(f"B={B}: eager {eager:.3f} ms, graph replay {graph:.3f} ms, bitwise equal {same}", flush=True)
