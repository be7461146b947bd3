#!/usr/bin/env python3
"""What the data-parallel machinery costs a B = 128 step when there is nothing to move (one-rank RCCL group):
(a) plain trainer, (b) events recorded inside the backward but no collective issued, (c) the full reducer (7 all-reduces on the
side stream behind the events), (d) the reducer without overlap (all collectives after the backward)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy
from vita_amd.trainer import PolicyTrainer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = default_config("hook_package")
cfg.policy.flow_matcher.name = "conditional"
S, A = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
B = 128
g = torch.Generator().manual_seed(1)
ncam = len(cfg.task.image_keys)
batch = {cfg.task.state_key: torch.randn(B, 1, S, generator=g).to(dev),
         "observation.features": torch.relu(torch.randn(B, 1, 512 * ncam, generator=g)).to(dev),
         cfg.task.action_key: torch.randn(B, 16, A, generator=g).to(dev)}
t = torch.rand(B, generator=g).to(dev)

def run(name, rehearse, overlap=True, no_collectives=False):
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats)
    tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100000, rehearse_collectives=rehearse, overlap=overlap)
    if no_collectives:
        tr.reducer.reduce = lambda i, ev=None: None
        tr.reducer.finish = lambda: 1.0
    tr.step_counter = 1000
    for _ in range(10):
        tr.train_step(batch, t=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        tr.train_step(batch, t=t)
    torch.cuda.synchronize()
    print(json.dumps({"variant": name, "ms_per_step": round((time.perf_counter() - t0) * 10, 4), "buckets": len(tr.plan)}), flush=True)

run("plain", False)
run("events_only", True, no_collectives=True)
run("reducer_overlap", True)
run("reducer_no_overlap", True, overlap=False)
run("plain_again", False)
dist.destroy_process_group()
