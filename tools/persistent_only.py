"""N persistent solves (B=4096, K=6, hook_package flow net) and nothing else: the program rocprofv3's counter passes run."""
import sys

import torch

sys.path.insert(0, ".")
from vita_amd import ops  # noqa: E402
from vita_amd.synthetic import make_state_dict  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fw = ops.FlowNetHandle(make_state_dict(533, 21, seed=0), dev)
x0 = torch.randn(4096, 512, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(n):
    ops.ode_sample(fw, x0, 6)
torch.cuda.synchronize()
print("status", ops.ode_sample_status(fw, 4096, 6, dev))
