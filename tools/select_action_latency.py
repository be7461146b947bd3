#!/usr/bin/env python3
"""End-to-end latency of VitaPolicy.select_action at eval_n_envs = 25 (vita_policy.py:210-220): a re-solve on every call
(the action queue is emptied each time), host time and wall time per call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy

dev = torch.device("cuda:0")
cfg = default_config("hook_package")
S, A = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.1] * S, "std": [0.9] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.2] * A}}
torch.manual_seed(0)
pol = VitaPolicy(cfg, stats).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ncam = len(cfg.task.image_keys)
batch = {cfg.task.state_key: torch.randn(n, S, device=dev), "observation.features": torch.relu(torch.randn(n, 512 * ncam, device=dev))}
for on in (True, False):
    ops.set_persistent_sampler(on)
    for _ in range(20):
        pol.reset(); pol.select_action(batch)
    torch.cuda.synchronize()
    hs = []
    t0 = time.perf_counter()
    for _ in range(200):
        pol.reset()
        h0 = time.perf_counter()
        a = pol.select_action(batch)
        hs.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    hs.sort()
    # latency of ONE call on an idle GPU: enqueue + wait for the action
    lat = []
    for _ in range(50):
        pol.reset()
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        a = pol.select_action(batch)
        a.cpu()
        lat.append(time.perf_counter() - h0)
    lat.sort()
    print(json.dumps({"n_envs": n, "resident": on, "back_to_back_ms_per_call": round(wall, 4), "host_ms_per_call_median": round(hs[100] * 1e3, 4),
                      "single_call_to_host_ms_median": round(lat[25] * 1e3, 4)}), flush=True)
ops.set_persistent_sampler(True)
