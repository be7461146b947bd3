"""cProfile of VitaPolicy.select_action at 25 envs (host-side cost around the one resident launch)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy
dev = torch.device("cuda:0")
cfg = default_config("hook_package")
S, A = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.1] * S, "std": [0.9] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.2] * A}}
torch.manual_seed(0)
pol = VitaPolicy(cfg, stats).to(dev)
ncam = len(cfg.task.image_keys)
batch = {cfg.task.state_key: torch.randn(25, S, device=dev), "observation.features": torch.relu(torch.randn(25, 512 * ncam, device=dev))}
for _ in range(20):
    pol.reset(); pol.select_action(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    pol.reset()
    pol.select_action(batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
