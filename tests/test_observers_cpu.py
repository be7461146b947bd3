"""vita_amd.observers against the reference's own observer code (tests/golden/observer.safetensors, tools/gen_golden_observer.py:
BaseObserver.get_images and ResNetObserver._observe_flattened/_observe_tokenized run around this repository's backbone).  The
ResNet18 definition itself restates torchvision, which is absent here: UNPINNED (its shapes, strides and names are checked)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from observer_fixture import backbone_checksum, seeded_backbone
from vita_amd.observers import BaseObserver, FrozenBatchNorm2d, ResNet18Backbone, ResNetObserver

GOLD = load_file(os.path.join(os.path.dirname(__file__), "golden", "observer.safetensors"))
KEYS = ["cam_a", "cam_b"]


def _batch():
    return {k[3:]: v for k, v in GOLD.items() if k.startswith("in/")}


@pytest.mark.parametrize("tag,rs,cs", [("resize", (48, 64), (40, 52)), ("same_size", (40, 56), (33, 41))])
def test_get_images_matches_reference(tag, rs, cs):
    obs = BaseObserver("observation.state", KEYS, rs, cs)
    torch.manual_seed(11)
    got = obs.get_images(_batch())
    want = GOLD[f"get_images/{tag}"]
    assert got.shape == want.shape
    # the crop position is the reference's two torch.randint draws: a shifted crop would differ by O(1)
    torch.testing.assert_close(got, want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("tok", [False, True])
def test_observe_matches_reference(tok):
    torch.set_num_threads(1)
    bb = seeded_backbone()
    assert torch.allclose(backbone_checksum(bb), GOLD["backbone_checksum"], rtol=1e-12, atol=0), "seeded backbone differs from the fixture's"
    obs = ResNetObserver("observation.state", KEYS, (48, 64), (40, 52), state_dim=10, tokenize=tok)
    obs.backbone.load_state_dict(bb.state_dict())
    if tok:
        obs.state_projector.load_state_dict({"weight": GOLD["state_projector/weight"], "bias": GOLD["state_projector/bias"]})
    torch.manual_seed(17)
    with torch.no_grad():
        got = obs(_batch())
    want = GOLD[f"observe/{'tokens' if tok else 'flat'}"]
    assert got.shape == want.shape == ((3, 1 + 2 * 2 * 9, 512) if tok else (3, 10 + 512 * 2 * 2))
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_backbone_names_and_shapes_are_torchvisions():
    bb = ResNet18Backbone()
    sd = bb.state_dict()
    assert sd["conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1) and "layer2.0.downsample.1.running_var" in sd
    assert "layer1.0.downsample.0.weight" not in sd and sd["layer4.1.conv2.weight"].shape == (512, 512, 3, 3)
    assert not any(k.startswith("fc.") for k in sd)
    # frozen norms hold buffers only: the trainable tensors are the 20 convolutions
    assert sum(1 for _ in bb.parameters()) == 20 and sum(p.numel() for p in bb.parameters()) == 11_166_912
    y = bb(torch.zeros(2, 3, 224, 308))["feature_map"]
    assert y.shape == (2, 512, 7, 10)


def test_frozen_batchnorm_is_eval_batchnorm():
    torch.manual_seed(0)
    f, ref = FrozenBatchNorm2d(6), torch.nn.BatchNorm2d(6).eval()
    with torch.no_grad():
        for name in ("weight", "bias", "running_mean"):
            getattr(f, name).normal_()
            getattr(ref, name).copy_(getattr(f, name))
        f.running_var.uniform_(0.5, 2.0)
        ref.running_var.copy_(f.running_var)
    x = torch.randn(3, 6, 5, 7)
    torch.testing.assert_close(f(x), ref(x), rtol=1e-5, atol=1e-6)


def test_load_torchvision_state_dict_ignores_head():
    obs = ResNetObserver(state_dim=4)
    sd = {k: torch.full_like(v, 0.5) for k, v in obs.backbone.state_dict().items()}
    sd["fc.weight"], sd["fc.bias"], sd["bn1.num_batches_tracked"] = torch.zeros(1000, 512), torch.zeros(1000), torch.tensor(0)
    obs.load_backbone_state_dict(sd)
    assert float(obs.backbone.layer3[1].bn2.running_var[0]) == 0.5


def test_backbone_against_an_independent_resnet18():
    """torchvision is absent, so the ResNet18 definition cannot be pinned to it; `transformers` ships an independent implementation of the
    same network (ResNetModel with basic layers — the class torchvision's resnet18 weights are converted into as microsoft/resnet-18).
    With the same tensors, in eval mode (BatchNorm on running statistics == FrozenBatchNorm2d), both must give the same layer4 map."""
    tr = pytest.importorskip("transformers")
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2], layer_type="basic",
                          hidden_act="relu", downsample_in_first_stage=False)
    ref = tr.ResNetModel(cfg).eval()
    bb = seeded_backbone()
    sd = bb.state_dict()

    def put(conv_layer, conv_key, bn_key):
        conv_layer.convolution.weight.data.copy_(sd[conv_key + ".weight"])
        for name in ("weight", "bias", "running_mean", "running_var"):
            getattr(conv_layer.normalization, name).data.copy_(sd[f"{bn_key}.{name}"])

    put(ref.embedder.embedder, "conv1", "bn1")
    for i, stage in enumerate(ref.encoder.stages):
        for j, layer in enumerate(stage.layers):
            pre = f"layer{i + 1}.{j}"
            put(layer.layer[0], pre + ".conv1", pre + ".bn1")
            put(layer.layer[1], pre + ".conv2", pre + ".bn2")
            if pre + ".downsample.0.weight" in sd:
                put(layer.shortcut, pre + ".downsample.0", pre + ".downsample.1")
            else:
                assert not hasattr(layer.shortcut, "convolution")
    assert sum(p.numel() for n, p in ref.named_parameters() if "normalization" not in n) == sum(p.numel() for p in bb.parameters())
    x = torch.rand(2, 3, 84, 100, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = ref(x).last_hidden_state
        got = bb(x)["feature_map"]
    assert got.shape == want.shape == (2, 512, 3, 4)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
