#!/usr/bin/env python3
"""What the ResNet18 observer (a torch module, MIOpen convolutions) costs beside the fused step: one drop-in training step
(compute_loss + backward) at hook_package shapes from camera frames vs from precomputed features.  GPU only."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd.config import default_config  # noqa: E402
from vita_amd.policy import VitaPolicy  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B = int(os.environ.get("TB", "128"))
    # (torch.backends.cudnn.benchmark = True — MIOpen's exhaustive find — ran for more than three silent minutes at these shapes and was
    # not measured; OBS_CHANNELS_LAST=1 measured 32.0 ms against 26.4: the hand-written passes and MIOpen's Winograd kernels are NCHW)
    cfg = default_config("hook_package")
    cfg.policy.flow_matcher.name = "conditional"
    s, a = cfg.task.state_dim, cfg.task.action_dim
    stats = {cfg.task.state_key: {"mean": [0.0] * s, "std": [1.0] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [1.0] * a}}
    out = {"B": B}
    phase = os.environ.get("PHASE", "all")       # "train": only the training step from frames (for a per-kernel profile)
    for name, observer in ((("features", None),) if phase == "all" else ()) + (("frames_resnet18", "resnet18"),):
        torch.manual_seed(0)
        pol = VitaPolicy(cfg, stats, observer=observer).to(DEV).train()
        if observer is not None and os.environ.get("OBS_CHANNELS_LAST") == "1":
            pol.observer.backbone.to(memory_format=torch.channels_last)
        batch = {cfg.task.state_key: torch.randn(B, 1, s, device=DEV), cfg.task.action_key: torch.randn(B, 16, a, device=DEV)}
        if observer is None:
            batch["observation.features"] = torch.randn(B, 1, 512, device=DEV).relu()
        else:
            batch[cfg.task.image_keys[0]] = torch.rand(B, 1, 3, 240, 320, device=DEV)
        t = torch.rand(B, device=DEV)

        def step():
            for p in pol.parameters():
                p.grad = None
            loss, _ = pol.compute_loss(batch, t=t)
            loss.backward()
        out[name + "_step_ms"] = round(timed(step, n=20 if phase == "train" else 10), 3)
        if phase == "train":
            print(json.dumps(out))
            return
        if observer is not None:
            with torch.no_grad():
                out["observer_fwd_ms"] = round(timed(lambda: pol.observer(batch)), 3)
            pol.eval()
            out["generate_actions_ms"] = round(timed(lambda: pol.generate_actions(batch)), 3)
    # the fused tail alone at the stem's shape: 8 B per element forward (read x, write y), 12 B backward (read dy, y; write dx)
    from vita_amd import ops
    x = torch.randn(B, 64, 112, 154, device=DEV, requires_grad=True)
    sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
    with torch.no_grad():
        ms = timed(lambda: ops.channel_affine_relu(x, sc, sh, None, True), n=20)
    out["tail_fwd_ms"], out["tail_fwd_GBps"] = round(ms, 4), round(x.numel() * 8 / ms / 1e6, 1)
    y = ops.channel_affine_relu(x, sc, sh, None, True)
    dy = torch.randn_like(y)
    ms = timed(lambda: torch.autograd.grad(y, x, dy, retain_graph=True), n=20)
    out["tail_bwd_ms"], out["tail_bwd_GBps"] = round(ms, 4), round(x.numel() * 12 / ms / 1e6, 1)
    out["torch_epilogue"] = os.environ.get("VITA_OBSERVER_TORCH_EPILOGUE", "0") == "1"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
