#!/usr/bin/env python3
"""The split-K reduction finished inside the few-row GEMM kernels (vita_splitk_fused.h: the last workgroup to arrive at a tile
sums the partials in split order) against the two-pass path (splitk_reduce_kernel): run with VITA_NO_FUSED_SPLITK=1 and =0;
`--save f` writes the outputs, `--compare f` checks them bitwise.  Covers the training step (losses + every gradient, B = 128
and a ragged 77) and the per-launch sampler at few-row batches (65 < B <= 512), several calls in a row (the tile counters
must be back at zero after every launch)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy

dev = torch.device("cuda:0")
cfg = default_config("hook_package")
cfg.policy.flow_matcher.name = "conditional"
s, a = cfg.task.state_dim, cfg.task.action_dim
stats = {cfg.task.state_key: {"mean": [0.0] * s, "std": [1.0] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [1.0] * a}}
torch.manual_seed(0)
pol = VitaPolicy(cfg, stats).to(dev)
pol.flatten_parameters()
pol.train()
outs = {}
ncam = len(cfg.task.image_keys)
for B in (128, 77, 128):
    g = torch.Generator().manual_seed(B)
    batch = {cfg.task.state_key: torch.randn(B, 1, s, generator=g).to(dev),
             "observation.features": torch.relu(torch.randn(B, 1, 512 * ncam, generator=g)).to(dev),
             cfg.task.action_key: torch.randn(B, 16, a, generator=g).to(dev)}
    t = torch.rand(B, generator=g).to(dev)
    batch = pol.normalize_targets(pol.normalize_inputs(batch))
    for rep in range(2):
        losses = pol.loss_and_grads(batch, t=t)
        torch.cuda.synchronize()
        outs[f"train{B}/{len(outs)}/losses"] = losses.cpu().clone()
        outs[f"train{B}/{len(outs)}/grads"] = pol._flat_grad.cpu().clone()
fw, dw, _ = pol._infer_handles()
for B in (65, 128, 200, 333, 512):
    x0 = torch.randn(B, 512, generator=torch.Generator().manual_seed(1000 + B)).to(dev)
    for rep in range(2):
        z = ops.ode_sample(fw, x0, 6)
        torch.cuda.synchronize()
        outs[f"sample{B}/{rep}"] = z.cpu().clone()
    assert torch.equal(outs[f"sample{B}/0"], outs[f"sample{B}/1"])
print(json.dumps({"fused": os.environ.get("VITA_NO_FUSED_SPLITK", "0") != "1", "n": len(outs)}))
if "--save" in sys.argv:
    torch.save(outs, sys.argv[sys.argv.index("--save") + 1])
if "--compare" in sys.argv:
    ref = torch.load(sys.argv[sys.argv.index("--compare") + 1], weights_only=True)
    bad = [k for k in outs if not torch.equal(outs[k], ref[k])]
    print(json.dumps({"bitwise_equal": not bad, "differ": bad}))
    if bad:
        sys.exit(1)
