#!/usr/bin/env python3
"""Batch assembly on the device-resident replay buffer: time per batch and achieved HBM GB/s (bytes of the batch read + written)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vita_amd.data import DeviceReplayDataset

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
lengths = [400] * 50
N = sum(lengths)
H, W = 240, 320
arrays = {"observation.state": rng.standard_normal((N, 21)).astype(np.float32),
          "observation.features": rng.standard_normal((N, 512)).astype(np.float32),
          "action": rng.standard_normal((N, 21)).astype(np.float32),
          "cam": rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)}
fps = 25
def run(keys, B, image_keys=()):
    dts = {"observation.state": [0.0], "observation.features": [0.0], "action": [i / fps for i in range(16)], "cam": [0.0]}
    ds = DeviceReplayDataset({k: arrays[k] for k in keys}, lengths, {k: dts[k] for k in keys}, fps, image_keys=image_keys, device=dev)
    idx = torch.from_numpy(rng.integers(0, N, size=B)).to(dev)
    for _ in range(5):
        ds.batch(idx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        b = ds.batch(idx)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    wr = sum(v.numel() * v.element_size() for k, v in b.items() if k in keys)
    rd = sum(v.numel() * (1 if k in image_keys else 4) for k, v in b.items() if k in keys)
    print(json.dumps({"keys": list(keys), "B": B, "ms_per_batch": round(ms, 4), "batch_MB": round(wr / 1e6, 2), "GB_per_s": round((wr + rd) / ms / 1e6, 1),
                      "frac_of_hbm_peak": round((wr + rd) / ms / 1e6 / 8000, 4), "resident_GB": round(ds.resident_bytes() / 1e9, 3)}), flush=True)
run(("observation.state", "observation.features", "action"), 128)
run(("observation.state", "observation.features", "action"), 4096)
run(("cam",), 128, image_keys=("cam",))
run(("cam",), 1024, image_keys=("cam",))
