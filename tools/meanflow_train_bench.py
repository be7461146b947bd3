#!/usr/bin/env python3
"""Training throughput of the MeanFlow policies (flow_matcher.name mean / improved_mean, simple_mean_flow_net, K = 1) on one GPU.
usage: python3 tools/meanflow_train_bench.py [B] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vita_amd.config import default_config  # noqa: E402
from vita_amd.policy import VitaPolicy  # noqa: E402
from vita_amd.trainer import PolicyTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
for name, disp in (("mean", 0.0), ("improved_mean", 0.0), ("mean", 0.5)):
    cfg = default_config("hook_package")
    cfg.policy.flow_matcher.name = name
    cfg.policy.flow_matcher.dispersive_loss_weight = disp
    cfg.policy.flow_matcher.num_sampling_steps = 1
    cfg.policy.flow_net.name = "simple_mean_flow_net"
    S, A = cfg.task.state_dim, cfg.task.action_dim
    stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats)
    with torch.no_grad():
        for k, p in pol.named_parameters():
            if "output_layer" in k:
                p.normal_(0.0, 0.02)
    tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100000)
    g = torch.Generator().manual_seed(100)
    ncam = len(cfg.task.image_keys)
    batch = {cfg.task.state_key: torch.randn(B, 1, S, generator=g).to(dev),
             "observation.features": torch.relu(torch.randn(B, 1, 512 * ncam, generator=g)).to(dev),
             cfg.task.action_key: torch.randn(B, 16, A, generator=g).to(dev)}
    smp = torch.sigmoid(torch.randn(B, 2, generator=g) - 0.4)
    t, r = smp.max(1)[0].to(dev), smp.min(1)[0].to(dev)
    tr.step_counter = 1000
    for _ in range(3):
        tr.train_step(batch, t=t, pair_i=r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(batch, t=t, pair_i=r)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name}{' + dispersive' if disp else ''}: B={B} ms_per_step={el * 1e3 / steps:.3f} samples_per_s={B * steps / el:.1f} loss={tr.metrics()['loss']:.6f}")
    del tr, pol
