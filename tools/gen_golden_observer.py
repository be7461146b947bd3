#!/usr/bin/env python3
"""tests/golden/observer.safetensors: the reference's BaseObserver.get_images and ResNetObserver._observe_flattened /
_observe_tokenized (flare/policies/observers/{base,resnet}_observer.py) around vita_amd.observers.ResNet18Backbone, on seeded
inputs (build container only).  torchvision is absent: it is stubbed for import resolution only, the reference ResNetObserver is
created without __init__ (which would build torchvision's resnet18) and given this repository's backbone — so the fixture pins the
observer's own logic (stack / resize / crop draw / pool / concat / tokens), not the ResNet18 definition."""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
from safetensors.torch import save_file

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
from vita_amd.observers import ResNet18Backbone, ResNetObserver  # noqa: E402


class _Absent:
    def __init__(self, *a, **k):
        raise RuntimeError("stand-in for an absent library: must not be used")


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


tv = stub("torchvision")
stub("torchvision.models"); stub("torchvision.models._utils", IntermediateLayerGetter=_Absent)
stub("torchvision.ops"); stub("torchvision.ops.misc", FrozenBatchNorm2d=_Absent)
# the reference's package __init__ files pull in libraries this image lacks (diffusers, ...): register the packages as bare
# namespaces so that only the two observer files themselves are executed
for pkg in ("flare", "flare.policies", "flare.policies.observers"):
    stub(pkg).__path__ = [os.path.join(REF, *pkg.split("."))]
from flare.policies.observers.base_observer import BaseObserver as RefBase  # noqa: E402
from flare.policies.observers.resnet_observer import ResNetObserver as RefResNet  # noqa: E402


def main():
    torch.set_num_threads(1)
    out = {}
    g = torch.Generator().manual_seed(0)
    b, s, keys, S = 3, 2, ["cam_a", "cam_b"], 5
    batch = {"observation.state": torch.randn(b, s, S, generator=g)}
    for k in keys:
        batch[k] = torch.rand(b, s, 3, 40, 56, generator=g)
    for k, v in batch.items():
        out[f"in/{k}"] = v
    # get_images: resize + crop draw, and the no-resize case
    for tag, (rs, cs) in {"resize": ((48, 64), (40, 52)), "same_size": ((40, 56), (33, 41))}.items():
        ref = RefBase(state_key="observation.state", image_keys=keys, resize_shape=rs, crop_shape=cs)
        torch.manual_seed(11)
        out[f"get_images/{tag}"] = ref.get_images(batch).contiguous()
    # observe, flattened and tokenised, around this repository's backbone
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from observer_fixture import backbone_checksum, seeded_backbone
    bb = seeded_backbone()
    out["backbone_checksum"] = backbone_checksum(bb)
    for tok in (False, True):
        ref = object.__new__(RefResNet)
        nn.Module.__init__(ref)
        ref.state_key, ref.image_keys, ref.resize_shape, ref.crop_shape = "observation.state", keys, (48, 64), (40, 52)
        ref.tokenize, ref.state_dim = tok, S * s
        ref.backbone = bb
        if tok:
            ref.spatial_pool = nn.AdaptiveAvgPool2d((3, 3))
            torch.manual_seed(5)
            ref.state_projector = nn.Linear(S * s, 512)
            out["state_projector/weight"], out["state_projector/bias"] = ref.state_projector.weight.detach().clone(), ref.state_projector.bias.detach().clone()
        else:
            ref.pool = nn.AdaptiveAvgPool2d((1, 1))
        torch.manual_seed(17)
        with torch.no_grad():
            y = ref.observe(batch)
        out[f"observe/{'tokens' if tok else 'flat'}"] = y.contiguous()
        mine = ResNetObserver("observation.state", keys, (48, 64), (40, 52), state_dim=S * s, tokenize=tok)
        mine.backbone.load_state_dict(bb.state_dict())
        if tok:
            mine.state_projector.load_state_dict(ref.state_projector.state_dict())
        torch.manual_seed(17)
        with torch.no_grad():
            y2 = mine(batch)
        assert y2.shape == y.shape and torch.equal(y2, y), (y2 - y).abs().max()
    save_file(out, os.path.join(REPO, "tests", "golden", "observer.safetensors"))
    print("ok", len(out), "tensors", sum(v.numel() * v.element_size() for v in out.values()) / 1e6, "MB")


if __name__ == "__main__":
    main()
