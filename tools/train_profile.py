#!/usr/bin/env python3
"""Training leg of bench.py alone (for rocprofv3 --kernel-trace --stats): N train steps at B rows on one GPU.
usage: python3 tools/train_profile.py [B] [steps] [f32|bf16|bf16_all]"""
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
cfg = default_config("hook_package")
cfg.policy.flow_matcher.name = "conditional"
cfg.policy.gemm_dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
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
tts = torch.rand(64, B, generator=g).to(dev)
tr.step_counter = 1000
if os.environ.get("VITA_PROFILE_STREAM"):      # run the step on a non-default torch stream
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
for i in range(3):
    tr.train_step(batch, t=tts[i % 64])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.train_step(batch, t=tts[i % 64])
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"B={B} steps={steps} ms_per_step={el * 1e3 / steps:.3f} samples_per_s={B * steps / el:.1f} loss={tr.metrics()['loss']:.6f}")
# host-side enqueue time of one step (GPU idle at the start): if it is close to ms_per_step the step is launch-rate bound
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.train_step(batch, t=tts[0])
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host enqueue time of one step: {host * 1e3:.3f} ms")
