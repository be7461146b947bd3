"""Observer front-end: camera frames + state -> the feature rows the hot path starts from.

`ResNetObserver` mirrors flare/policies/observers/resnet_observer.py:11-97 over base_observer.py:7-44: stack the cameras, bilinear
resize to `resize_shape`, ONE random crop per call (two torch.randint draws, training and evaluation alike), ResNet18 up to layer4
with frozen batch-norm statistics, average pool, and `[state | camera features]` per row (or the tokenised form).  It is a torch
module that plugs into `VitaPolicy(observer=...)`: its convolutions are MIOpen's; everything between them — frozen batch norm,
shortcut add, ReLU, and the stem's max pool — is one hand-written pass per convolution on the device (vita_channel_affine_relu_f32,
vita_affine_relu_maxpool3s2_f32 and their backwards, include/vita_hip.h; bitwise the unfused torch arithmetic, VITA_OBSERVER_TORCH_EPILOGUE=1 runs the torch ops instead; CPU tensors
always take the torch ops, which is how the module's logic is pinned against the reference in the build container).  Its
output carries autograd, and the fused step returns d loss / d features (vita_compute_loss_f32's d_obs_features), so the
backbone trains on the drop-in path as in the reference (vita_policy.py:222-229 puts it in the same AdamW group).

Parity: the observer's own logic (image stacking / resize / crop / pooling / concatenation / tokenisation) is pinned by running the
reference's BaseObserver.get_images and ResNetObserver._observe_* around this file's backbone (tests/golden/observer.safetensors).
The ResNet18 definition itself and FrozenBatchNorm2d restate torchvision (absent from this image): UNPINNED against it, cross-checked
against transformers' independent ResNetModel (basic layers) on the same tensors (tests/test_observers_cpu.py).  Parameter and buffer
names are torchvision's behind `backbone.`, so a reference checkpoint's `observer.*` tensors load; ImageNet weights are not
downloaded here — pass a local torchvision state dict to `load_backbone_state_dict`."""
from __future__ import annotations

import os
from typing import Mapping, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class FrozenBatchNorm2d(nn.Module):
    """torchvision.ops.misc.FrozenBatchNorm2d (restated): statistics and affine are buffers; y = x * scale + shift with
    scale = weight * rsqrt(running_var + eps), shift = bias - running_mean * scale."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x: torch.Tensor, shortcut: torch.Tensor | None = None, relu: bool = False) -> torch.Tensor:
        """act(x * scale + shift [+ shortcut]): the norm together with what follows it in a residual block."""
        scale, shift = self.scale_shift()
        if x.is_cuda and os.environ.get("VITA_OBSERVER_TORCH_EPILOGUE", "0") != "1":
            from . import ops
            return ops.channel_affine_relu(x, scale, shift, shortcut, relu)
        y = x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
        if shortcut is not None:
            y = y + shortcut
        return F.relu(y) if relu else y


class _BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock (restated): conv3x3-bn-relu-conv3x3-bn, identity or 1x1-conv shortcut, relu."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = FrozenBatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), FrozenBatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(out), shortcut=idt, relu=True)


class ResNet18Backbone(nn.Module):
    """resnet18(norm_layer=FrozenBatchNorm2d) up to layer4 — what IntermediateLayerGetter(backbone, {"layer4": "feature_map"})
    keeps (resnet_observer.py:40-44); module names are torchvision's."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths, cin = (64, 128, 256, 512), 64
        for i, c in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(_BasicBlock(cin, c, stride), _BasicBlock(c, c, 1)))
            cin = c
        for m in self.modules():        # torchvision's init for the convolutions; the frozen norms start as identity
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.conv1(x)
        if x.is_cuda and os.environ.get("VITA_OBSERVER_TORCH_EPILOGUE", "0") != "1":
            from . import ops
            x = ops.affine_relu_maxpool(x, *self.bn1.scale_shift())          # norm + ReLU + 3x3/2 max pool: one pass
        else:
            x = self.maxpool(self.bn1(x, relu=True))
        return {"feature_map": self.layer4(self.layer3(self.layer2(self.layer1(x))))}


class BaseObserver(nn.Module):
    """base_observer.py:7-44."""

    def __init__(self, state_key: str = "observation.state", image_keys: Sequence[str] = ("observation.image",),
                 resize_shape=(240, 320), crop_shape=(224, 308)):
        super().__init__()
        self.state_key, self.image_keys = state_key, list(image_keys)
        self.resize_shape, self.crop_shape = tuple(resize_shape), tuple(crop_shape)

    def get_states(self, batch):
        return batch[self.state_key]

    def get_images(self, batch):
        """(b, s, c, h, w) per camera -> (b * s * n, c, ch, cw): resize when the size differs, then one crop position for the whole
        call, drawn with torch.randint (top first, then left) whether training or not (:24-36)."""
        images = torch.stack([batch[k] for k in self.image_keys], dim=2)            # b s n c h w
        images = images.reshape((-1,) + tuple(images.shape[3:]))
        if tuple(images.shape[-2:]) != self.resize_shape:
            images = F.interpolate(images, size=self.resize_shape, mode="bilinear", align_corners=False)
        ch, cw = self.crop_shape
        H, W = images.shape[-2:]
        top = torch.randint(0, H - ch + 1, (1,)).item()
        left = torch.randint(0, W - cw + 1, (1,)).item()
        return images[..., top:top + ch, left:left + cw]

    def observe(self, batch):
        raise NotImplementedError("observe() should be implemented.")

    def forward(self, batch):
        return self.observe(batch)


class ResNetObserver(BaseObserver):
    """resnet_observer.py:11-97.  tokenize=False: (b, state_dim * s + 512 * s * n) rows [state | pooled camera features];
    tokenize=True: (b, 1 + s * n * 9, 512) tokens (projected state first, then a 3 x 3 grid per image)."""

    def __init__(self, state_key: str = "observation.state", image_keys: Sequence[str] = ("observation.image",), resize_shape=(240, 320),
                 crop_shape=(224, 308), state_dim: int = 21, tokenize: bool = False):
        super().__init__(state_key, image_keys, resize_shape, crop_shape)
        self.tokenize, self.state_dim = tokenize, state_dim
        self.backbone = ResNet18Backbone()
        if tokenize:
            self.spatial_pool = nn.AdaptiveAvgPool2d((3, 3))
            self.state_projector = nn.Linear(state_dim, 512)
        else:
            self.pool = nn.AdaptiveAvgPool2d((1, 1))

    def load_backbone_state_dict(self, sd: Mapping[str, torch.Tensor], strict: bool = True):
        """A torchvision resnet18 state dict from a local file (`fc.*` and `num_batches_tracked` entries are ignored)."""
        keep = {k: v for k, v in sd.items() if not k.startswith("fc.") and not k.endswith("num_batches_tracked")}
        return self.backbone.load_state_dict(keep, strict=strict)

    def observe(self, batch):
        return self._observe_tokenized(batch) if self.tokenize else self._observe_flattened(batch)

    def _observe_flattened(self, batch):                                             # :80-97
        states = self.get_states(batch)
        b, s, n = states.shape[0], states.shape[1], len(self.image_keys)
        feats = self.pool(self.backbone(self.get_images(batch))["feature_map"])      # (b s n, 512, 1, 1)
        return torch.cat([states.flatten(start_dim=1), feats.reshape(b, s * n * feats.shape[1])], dim=1)

    def _observe_tokenized(self, batch):                                             # :57-78
        b, s, n = batch[self.state_key].shape[0], batch[self.image_keys[0]].shape[1], len(self.image_keys)
        fm = self.spatial_pool(self.backbone(self.get_images(batch))["feature_map"])  # (b s n, c, 3, 3)
        tokens = fm.flatten(2).transpose(1, 2).reshape(b, s * n * 9, fm.shape[1])
        state_tokens = self.state_projector(self.get_states(batch).flatten(start_dim=1)).unsqueeze(1)
        return torch.cat([state_tokens, tokens], dim=1)
