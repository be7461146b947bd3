"""Host enqueue time vs GPU time of one training step (is the step host-bound?).  python tools/host_vs_gpu.py [B]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy
from vita_amd.trainer import PolicyTrainer

TB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
cfg = default_config("hook_package")
cfg.policy.flow_matcher.name = "conditional"
S, A = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
torch.manual_seed(0)
pol = VitaPolicy(cfg, stats)
tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100000)
g = torch.Generator().manual_seed(1)
ncam = len(cfg.task.image_keys)
batch = {cfg.task.state_key: torch.randn(TB, 1, S, generator=g).to(dev),
         "observation.features": torch.relu(torch.randn(TB, 1, 512 * ncam, generator=g)).to(dev),
         cfg.task.action_key: torch.randn(TB, 16, A, generator=g).to(dev)}
tt = torch.rand(TB, generator=g).to(dev)
tr.step_counter = 1000
for _ in range(5):
    tr.train_step(batch, t=tt)
torch.cuda.synchronize()
# (a) host alone: enqueue one step on an idle GPU, time until the call returns, then until the GPU is done
hs, ws = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(batch, t=tt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(t1 - t0); ws.append(t2 - t0)
hs.sort(); ws.sort()
# (b) back to back
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    tr.train_step(batch, t=tt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print({"B": TB, "host_enqueue_ms_median": round(hs[10] * 1e3, 3), "enqueue_to_done_ms_median": round(ws[10] * 1e3, 3),
       "back_to_back_ms_per_step": round((t2 - t0) / 50 * 1e3, 3), "host_ms_per_step_back_to_back": round((t1 - t0) / 50 * 1e3, 3)})
