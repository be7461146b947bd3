"""The seeded ResNet18 backbone behind tests/golden/observer.safetensors (45 MB of weights are regenerated from the seed instead
of stored); shared by tools/gen_golden_observer.py and tests/test_observers_cpu.py."""
import torch


def seeded_backbone():
    from vita_amd.observers import ResNet18Backbone
    torch.manual_seed(3)
    bb = ResNet18Backbone()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():        # non-trivial frozen statistics
        for m in bb.modules():
            if hasattr(m, "running_var"):
                m.running_var.uniform_(0.5, 1.5, generator=g)
                m.running_mean.normal_(0, 0.1, generator=g)
                m.weight.uniform_(0.8, 1.2, generator=g)
                m.bias.normal_(0, 0.1, generator=g)
    return bb


def backbone_checksum(bb):
    return torch.stack([v.double().sum() for v in bb.state_dict().values()]).sum().reshape(1)
