#!/usr/bin/env python3
"""bf16-operand generate_actions at 1024..4096 rows (fold on / off via VITA_NO_LN_FOLD)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict
dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
bw = ops.Bf16Weights({k: v.to(dev) for k, v in sd.items()}, dev)
ob = sd["obs_encoder.bias"].to(dev)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (512, 1024, 1536, 2048, 3072, 4096):
    obs = torch.randn(B, 533).to(dev); out = torch.empty(B, 336, device=dev)
    print(json.dumps({"B": B, "fold_off": os.environ.get("VITA_NO_LN_FOLD", "0"), "ms": round(timed(lambda: ops.generate_actions_bf16(ob, fw, dw, bw, obs, 6, out=out)), 4)}), flush=True)
