"""Where a persistent solve's time goes: per workgroup and phase, 100 MHz clock stamps (phase start, wait over, tiles done,
published) -> mean / max of wait, compute and publish per phase kind."""
import json
import sys

import torch

sys.path.insert(0, ".")
from vita_amd import _lib, ops  # noqa: E402
from vita_amd.synthetic import make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
B, K, L = 4096, 6, 4
sd = make_state_dict(533, 21, seed=0)
fw = ops.FlowNetHandle(sd, dev)
x0 = torch.randn(B, 512, generator=torch.Generator().manual_seed(1)).to(dev)
PH = 2 + 2 * L
buf = torch.zeros(256 * K * PH * 6, dtype=torch.int64, device=dev)
lib = _lib.load()
BF16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
if BF16:
    dw, bw = ops.DecoderHandle(sd, dev), ops.Bf16Weights(sd, dev)
    obs = torch.randn(B, 533, generator=torch.Generator().manual_seed(2)).to(dev)
    ob = sd["obs_encoder.bias"].to(dev)
    run = lambda: ops.generate_actions_bf16(ob, fw, dw, bw, obs, K)   # noqa: E731
else:
    run = lambda: ops.ode_sample(fw, x0, K)   # noqa: E731
for _ in range(3):
    run()
lib.vita_debug_ode_stamps(buf.data_ptr())
run()
torch.cuda.synchronize()
lib.vita_debug_ode_stamps(None)
raw = buf.cpu().view(256, K * PH, 6).double()
s = raw[:, :, :4] / 100.0     # microseconds
t0 = s[:, 0, 0].min()
kinds = {"in_proj": [], "fc1": [], "fc2": [], "out_proj": []}
for p in range(K * PH):
    ph = p % PH
    kind = "in_proj" if ph == 0 else "out_proj" if ph == PH - 1 else ("fc1" if (ph - 1) % 2 == 0 else "fc2")
    kinds[kind].append(p)
out = {"total_us": float(s[:, -1, 3].max() - t0)}
out["shader_clock_GHz"] = round(((raw[:, -1, 5] - raw[:, 0, 4]) / (raw[:, -1, 3] - raw[:, 0, 0])).mean().item() * 0.1, 4)
for kind_, p_ in (("fc1_phase1", 1), ("fc2_phase2", 2)):
    out["clock_" + kind_] = round(((raw[:, p_, 5] - raw[:, p_, 4]) / (raw[:, p_, 3] - raw[:, p_, 0])).mean().item() * 0.1, 4)
for kind, ps in kinds.items():
    w = (s[:, ps, 1] - s[:, ps, 0])
    c = (s[:, ps, 2] - s[:, ps, 1])
    d = (s[:, ps, 3] - s[:, ps, 2])
    out[kind] = {"wait_mean": round(w.mean().item(), 2), "wait_max": round(w.max().item(), 2), "tile_mean": round(c.mean().item(), 2),
                 "tile_min": round(c.min().item(), 2), "tile_max": round(c.max().item(), 2), "drain_mean": round(d.mean().item(), 2),
                 "phase_mean": round((s[:, ps, 3] - s[:, ps, 0]).mean().item(), 2)}
print(json.dumps(out))
