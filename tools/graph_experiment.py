#!/usr/bin/env python3
"""Experiment: one whole train step (observer glue + fused loss step + optimizer) captured in a HIP graph and replayed,
against the same step enqueued eagerly.  The captured step bakes the learning rate and the step count in, so this times
the launch path only.  usage: python3 tools/graph_experiment.py [B] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vita_amd.config import default_config  # noqa: E402
from vita_amd.policy import VitaPolicy  # noqa: E402
from vita_amd.trainer import PolicyTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
cfg = default_config("hook_package")
cfg.policy.flow_matcher.name = "conditional"
S, A = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
torch.manual_seed(0)
pol = VitaPolicy(cfg, stats)
tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100000)
g = torch.Generator().manual_seed(100)
ncam = len(cfg.task.image_keys)
batch = {cfg.task.state_key: torch.randn(B, 1, S, generator=g).to(dev),
         "observation.features": torch.relu(torch.randn(B, 1, 512 * ncam, generator=g)).to(dev),
         cfg.task.action_key: torch.randn(B, 16, A, generator=g).to(dev)}
tt = torch.rand(B, generator=g).to(dev)
tr.step_counter = 1000
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
for i in range(3):
    tr.train_step(batch, t=tt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.train_step(batch, t=tt)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) * 1e3 / steps
print(f"eager: {eager:.3f} ms/step", flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    tr.train_step(batch, t=tt)
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    graph.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) * 1e3 / steps
print(f"graph replay: {rep:.3f} ms/step  (x{eager / rep:.2f})  loss={tr.metrics()['loss']:.6f}", flush=True)
