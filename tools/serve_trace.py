"""A short generate_actions loop at B = 25 for rocprofv3 --kernel-trace (which kernels a serving solve still launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vita_amd import ops
from vita_amd.synthetic import make_state_dict
dev = torch.device("cuda:0")
sd = make_state_dict(533, 21, seed=0)
fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
obs = torch.randn(25, 533).to(dev)
out = torch.empty(25, 336, device=dev)
for _ in range(30):
    ops.generate_actions(ow, ob, fw, dw, obs, 6, out=out)
torch.cuda.synchronize()
