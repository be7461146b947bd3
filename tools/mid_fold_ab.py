#!/usr/bin/env python3
"""generate_actions at mid batches: LayerNorm folded into the GEMM epilogues (split-K + folded second pass, round 3) against the
LayerNorm-kernel path.  Run once per setting: VITA_LN_FOLD_MIN_ROWS=1024 (default) / =3072 (round 2's rule) / VITA_NO_LN_FOLD=1."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
row = {"fold_min_rows": os.environ.get("VITA_LN_FOLD_MIN_ROWS", "1024"), "no_fold": os.environ.get("VITA_NO_LN_FOLD", "0")}
for B in (512, 1024, 1536, 2048, 2560, 3072, 4096):
    obs = torch.randn(B, 533, generator=torch.Generator().manual_seed(B)).to(dev)
    out = torch.empty(B, 336, device=dev)
    for _ in range(5):
        ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
    torch.cuda.synchronize()
    row[f"B{B}_ms"] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
print(json.dumps(row), flush=True)
