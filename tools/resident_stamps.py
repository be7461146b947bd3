"""Where a weights-resident solve's time goes: per workgroup and phase, 100 MHz clock stamps (phase start, wait over, A landed,
sums in LDS, published) -> mean of each segment per phase kind.  python tools/resident_stamps.py [B]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from vita_amd import _lib, ops  # noqa: E402
from vita_amd.synthetic import make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
K, L = 6, 4
sd = make_state_dict(533, 21, seed=0)
fw = ops.FlowNetHandle(sd, dev)
x0 = torch.randn(B, 512, generator=torch.Generator().manual_seed(1)).to(dev)
PH = 2 + 2 * L
TP = 3      # time-path phases ahead of the steps (K <= 32, TE = 256)
NPH = K * PH + TP
buf = torch.zeros(256 * NPH * 6, dtype=torch.int64, device=dev)
lib = _lib.load()
for _ in range(3):
    ops.ode_sample(fw, x0, K)
lib.vita_debug_ode_stamps(buf.data_ptr())
ops.ode_sample(fw, x0, K)
torch.cuda.synchronize()
lib.vita_debug_ode_stamps(None)
s = buf.cpu().view(256, NPH, 6).double() / 100.0     # microseconds
kinds = {"in_proj": [], "fc1": [], "fc2": [], "out_proj": []}
kinds["time_path"] = [1, 2]
for p in range(TP, NPH):
    ph = (p - TP) % PH
    kinds["in_proj" if ph == 0 else "out_proj" if ph == PH - 1 else ("fc1" if (ph - 1) % 2 == 0 else "fc2")].append(p)
out = {"B": B, "total_us": float(s[:, -1, 4].max() - s[:, 0, 0].min()), "first_phase_start_spread_us": float(s[:, 0, 0].max() - s[:, 0, 0].min())}
for kind, ps in kinds.items():
    seg = lambda a, b: round((s[:, ps, b] - s[:, ps, a]).mean().item(), 2)   # noqa: E731
    # skew: how long after the LAST workgroup published the previous phase a workgroup's wait ends
    last_pub = s[:, [p - 1 for p in ps], 4].max(dim=0).values
    out[kind] = {"wait": seg(0, 1), "a_landed": seg(1, 2), "ln": seg(2, 5), "mfma_reduce": seg(5, 3), "epilogue_publish": seg(3, 4), "phase": seg(0, 4),
                 "wait_end_after_last_publish": round((s[:, ps, 1] - last_pub[None, :]).mean().item(), 2),
                 "publish_spread": round((s[:, ps, 4].max(dim=0).values - s[:, ps, 4].min(dim=0).values).mean().item(), 2)}
print(json.dumps(out))
