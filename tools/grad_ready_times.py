"""When each gradient-ready event of the fused backward fires inside a B = 128 training step (the window a data-parallel
all-reduce of that bucket has before the step's end), and how many bytes each bucket holds."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy

TB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
cfg = default_config("hook_package")
cfg.policy.flow_matcher.name = "conditional"
S, A = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
torch.manual_seed(0)
pol = VitaPolicy(cfg, stats).to(dev)
pol.flatten_parameters()
g = torch.Generator().manual_seed(1)
ncam = len(cfg.task.image_keys)
batch = {cfg.task.state_key: torch.randn(TB, 1, S, generator=g).to(dev),
         "observation.features": torch.relu(torch.randn(TB, 1, 512 * ncam, generator=g)).to(dev),
         cfg.task.action_key: torch.randn(TB, 16, A, generator=g).to(dev)}
tt = torch.rand(TB, generator=g).to(dev)
plan = pol.reduce_plan()
nev = 1 + max(ev for _, _, ev in plan)
events = [torch.cuda.Event(enable_timing=True) for _ in range(nev)]
for e in events:
    e.record()
start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pol.train()
for _ in range(5):
    pol.loss_and_grads(batch, t=tt, grad_ready_events=events)
torch.cuda.synchronize()
start.record()
pol.loss_and_grads(batch, t=tt, grad_ready_events=events)
end.record()
torch.cuda.synchronize()
total = start.elapsed_time(end)
rows = []
for lo, hi, ev in plan:
    rows.append({"event": ev, "MB": round((hi - lo) * 4 / 1e6, 2), "fires_at_ms": round(start.elapsed_time(events[ev]), 3),
                 "window_to_end_ms": round(total - start.elapsed_time(events[ev]), 3)})
print(json.dumps({"B": TB, "fwd_bwd_ms": round(total, 3), "buckets_in_reduce_order": rows}))
