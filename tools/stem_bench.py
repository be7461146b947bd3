#!/usr/bin/env python3
"""The stem pass (frozen norm + ReLU + max pool) alone at 128 x 64 x 112 x 154: forward and backward time and algorithmic GB/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, n=20, w=3):
    for _ in range(w):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    W = int(os.environ.get("STEM_W", "154"))           # 154: hook_package (308-wide crops); 112: the 224-pixel crops
    x = torch.randn(128, 64, 112, W, device=DEV, requires_grad=True)
    sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1
    n = x.numel()
    with torch.no_grad():
        f = timed(lambda: ops.affine_relu_maxpool(x, sc, sh))
    y = ops.affine_relu_maxpool(x, sc, sh)
    dy = torch.randn_like(y)
    b = timed(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
    nb = 4 * n + 5 * (n // 4)
    print(json.dumps({"W": W, "fwd_ms": round(f, 4), "fwd_GBps": round(nb / f / 1e6, 1), "bwd_ms": round(b, 4), "bwd_GBps": round(nb / b / 1e6, 1)}))


if __name__ == "__main__":
    main()
