#!/usr/bin/env python3
"""B=128 training step of the default policy: time per step, and the same with VITA_NO_FUSED_SPLITK=1 (child process)."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from vita_amd.config import default_config
    from vita_amd.policy import VitaPolicy
    from vita_amd.trainer import PolicyTrainer
    dev = torch.device("cuda:0")
    cfg = default_config("hook_package")
    cfg.policy.flow_matcher.name = os.environ.get("FM", "conditional")     # identity pairing (the OT draw is host work)
    s, a = cfg.task.state_dim, cfg.task.action_dim
    stats = {cfg.task.state_key: {"mean": [0.0] * s, "std": [1.0] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [1.0] * a}}
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats).to(dev)
    tr = PolicyTrainer(cfg, pol, dev, num_training_steps=1000)
    B = int(os.environ.get("TB", "128"))
    g = torch.Generator().manual_seed(1)
    batch = {cfg.task.state_key: torch.randn(B, 1, s, generator=g).to(dev),
             "observation.features": torch.relu(torch.randn(B, 1, 512 * len(cfg.task.image_keys), generator=g)).to(dev),
             cfg.task.action_key: torch.randn(B, 16, a, generator=g).to(dev)}
    t = torch.rand(B, generator=g).to(dev)
    for _ in range(10):
        tr.train_step(batch, t=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        tr.train_step(batch, t=t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print(json.dumps({"B": B, "fused_splitk": os.environ.get("VITA_NO_FUSED_SPLITK", "0") != "1", "ms_per_step": round(ms, 4),
                      "loss": round(float(tr.metrics()["loss"]), 6)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        variants = ({}, {"VITA_NO_FUSED_SPLITK": "1"}, {}, {"VITA_NO_FUSED_SPLITK": "1"})
        for env in variants:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
