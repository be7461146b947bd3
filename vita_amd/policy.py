"""VitaPolicy — drop-in for flare.policies.vita.vita_policy.VitaPolicy on the vision-latent -> action-latent
flow-matching path (same constructor, methods, metric keys and state-dict keys).

PyTorch holds the parameters (device memory, state_dict, optimizer compatibility); every FLOP of
generate_actions / select_action / compute_loss runs in libvita_hip.so through the C ABI:
  * inference:  vita_generate_actions_f32  (obs_encoder -> K Euler steps -> decoder, one call)
  * training:   vita_compute_loss_f32      (forward + hand-written backward, one call)
There is no CPU or torch fallback: on a CPU tensor the ops raise VitaHipError.

The observer sits in front of that path as a torch module: `observer="resnet18"` builds
vita_amd.observers.ResNetObserver from the config as vita_policy.py:43-50 does (camera frames in, backbone trained through
d loss / d features); any nn.Module mapping the batch to (B, O) features can be passed instead; the default FeatureObserver
consumes precomputed camera latents (batch["observation.features"], 512 per camera) + the state, in the reference's concat
order (resnet_observer.py:80-97: state first).
"""
from __future__ import annotations

import math
import os
import time
from collections import deque
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .flow import MeanFlowMatcher, get_flow_matcher

MEAN_FLOW_NET_NAME = "simple_mean_flow_net"          # vita_policy.py:86-87
FLOW_NET_NAMES = ("simple_flow_net", "simple")  # vita.yaml:28 ships "simple", vita_policy.py:84 accepts "simple_flow_net"


# ----------------------------------------------------------------------------------------------------
# normalisation (flare/utils/normalize.py) — same module/parameter names, HIP kernels for the arithmetic
# ----------------------------------------------------------------------------------------------------
def _stats_buffers(module: nn.Module, keys, stats) -> None:
    for key, stat in stats.items():
        if key not in keys:
            continue
        buf = nn.ParameterDict({
            "mean": nn.Parameter(torch.as_tensor(stat["mean"], dtype=torch.float32).clone(), requires_grad=False),
            "std": nn.Parameter(torch.as_tensor(stat["std"], dtype=torch.float32).clone(), requires_grad=False),
        })
        setattr(module, "buffer_" + key.replace(".", "_"), buf)


class Normalize(nn.Module):
    """normalize.py:25-46: (x - mean) / (std + 1e-8) per key."""
    _fn = staticmethod(ops.normalize)

    def __init__(self, keys, stats):
        super().__init__()
        self.keys = list(keys)
        _stats_buffers(self, self.keys, stats)

    @torch.no_grad()
    def forward(self, batch):
        batch = dict(batch)
        for key in self.keys:
            name = "buffer_" + key.replace(".", "_")
            if key not in batch or not hasattr(self, name):
                continue   # e.g. image keys when the batch carries precomputed camera latents
            buf = getattr(self, name)
            batch[key] = self._fn(batch[key], buf["mean"], buf["std"])
        return batch


class Unnormalize(Normalize):
    """normalize.py:49-70: x * std + mean per key."""
    _fn = staticmethod(ops.unnormalize)


# ----------------------------------------------------------------------------------------------------
# parameter containers with the reference's state-dict keys (never called: the kernels read the tensors)
# ----------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the arithmetic runs in libvita_hip.so (see VitaPolicy)")


def _mlp(in_f, hid_f, out_f):
    m = _Holder()
    m.fc1, m.fc2 = nn.Linear(in_f, hid_f), nn.Linear(hid_f, out_f)
    return m


def _xavier(mod: nn.Module):
    for m in mod.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)


def _orthogonal(mod: nn.Module):
    for m in mod.modules():
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            nn.init.orthogonal_(m.weight.data)
            m.bias.data.fill_(0.0)


def make_flow_net(D, H, L_, mlp_ratio, time_embed_dim=256) -> nn.Module:
    """SimpleFlowNet's parameters and init (simple_flow_net.py:52-97; note the Xavier pass re-initialises the
    zero-initialised modulators)."""
    net = _Holder()
    net.input_proj = nn.Linear(D, H)
    net.time_embed = nn.Sequential(nn.Identity(), nn.Linear(time_embed_dim, 4 * time_embed_dim), nn.Mish(),
                                   nn.Linear(4 * time_embed_dim, H))
    layers = []
    for _ in range(L_):
        layer = _Holder()
        layer.mlp = _mlp(H, int(H * mlp_ratio), H)
        layer.time_modulator = nn.Sequential(nn.SiLU(), nn.Linear(H, 3 * H))
        layers.append(layer)
    net.layers = nn.ModuleList(layers)
    net.norm = nn.LayerNorm(H)
    net.out_proj = nn.Linear(H, D)
    _xavier(net)
    nn.init.normal_(net.time_embed[1].weight, std=0.02)
    nn.init.normal_(net.time_embed[3].weight, std=0.02)
    return net


def make_mean_flow_net(D, H, L_, mlp_ratio) -> nn.Module:
    """SimpleMeanFlowNet's parameters, keys and init (simple_mean_flow_net.py:20-61: Xavier, zero biases, zero u/v heads)."""
    net = _Holder()
    net.t_embed = nn.Sequential(nn.Linear(1, H), nn.SiLU(), nn.Linear(H, H))
    net.h_embed = nn.Sequential(nn.Linear(1, H), nn.SiLU(), nn.Linear(H, H))
    net.input_layer = nn.Linear(D, H)
    dim = int(H * mlp_ratio)
    net.hidden_layers = nn.ModuleList([nn.Sequential(nn.Linear(H, dim), nn.SiLU(), nn.Dropout(p=0.0), nn.Linear(dim, H))
                                       for _ in range(L_)])
    net.u_output_layer = nn.Linear(H, D)
    net.v_output_layer = nn.Linear(H, D)
    for m in net.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0)
    for head in (net.u_output_layer, net.v_output_layer):
        nn.init.constant_(head.weight, 0)
        nn.init.constant_(head.bias, 0)
    return net


def make_action_encoder(kind, T, A, D, hidden, num_layers, variational: bool = False) -> nn.Module:
    """get_autoencoder (action_ae.py:330-374) / get_variational_autoencoder (action_vae.py:390-420): the variational CNN
    encoder differs only in latent_proj's width (2 * latent_dim: [mean | logvar], action_vae.py:213)."""
    enc = _Holder()
    if variational and kind != "cnn":
        if kind == "transformer":
            raise NotImplementedError("encoder_type 'transformer' is outside this build's hot path (SURVEY.md section 2 #3)")
        raise ValueError(f"Unknown encoder type: {kind}")   # action_vae.py:420
    if kind == "cnn":                                   # action_ae.py:282-316
        if T // (2 ** num_layers) < 1 or T % (2 ** num_layers):
            raise ValueError(f"cnn encoder: pred_horizon {T} must be a multiple of 2^num_layers ({num_layers})")
        mods, c = [], A
        for _ in range(num_layers):
            mods += [nn.Conv1d(c, hidden, kernel_size=5, stride=2, padding=2), nn.ReLU()]
            c = hidden
        enc.encoder = nn.Sequential(*mods)
        enc.latent_proj = nn.Linear(hidden * (T // (2 ** num_layers)), 2 * D if variational else D)
        _orthogonal(enc)
    elif kind == "simple":                              # action_ae.py:111-159
        ald = D // T
        enc.input_proj = nn.Linear(A, ald)
        enc.layers = nn.ModuleList([_mlp(ald, 4 * ald, ald) for _ in range(num_layers)])
        _xavier(enc)
    elif kind == "transformer":
        raise NotImplementedError("encoder_type 'transformer' is outside this build's hot path (SURVEY.md section 2 #3)")
    else:
        raise ValueError(f"Unknown encoder type: {kind}")   # action_ae.py:374
    return enc


def make_action_decoder(kind, T, A, D, hidden, num_layers) -> nn.Module:
    if kind != "simple":
        raise ValueError(f"Unknown encoder type: {kind}")   # sic, action_ae.py:386
    dec = _Holder()
    dec.input_proj = nn.Linear(D, hidden)                   # action_ae.py:239-253
    dec.output_proj = nn.Linear(hidden, T * A)
    dec.layers = nn.ModuleList([_mlp(hidden, hidden, hidden) for _ in range(num_layers)])
    _xavier(dec)
    return dec


class FeatureObserver(nn.Module):
    """Stand-in for ResNetObserver._observe_flattened (resnet_observer.py:80-97) when the camera latents are
    precomputed: cat([state.flatten(1), features.flatten(1)], dim=1).  Pure data movement."""

    def __init__(self, state_key: str, feature_key: str = "observation.features"):
        super().__init__()
        self.state_key, self.feature_key = state_key, feature_key

    def forward(self, batch):
        if self.feature_key not in batch:
            raise KeyError(f"batch has no '{self.feature_key}': pass precomputed camera latents, or construct VitaPolicy with "
                           "observer='resnet18' (camera frames in) or observer=<module producing (B, O) features>")
        return torch.cat([batch[self.state_key].flatten(start_dim=1), batch[self.feature_key].flatten(start_dim=1)], dim=1)


# ----------------------------------------------------------------------------------------------------
class _FusedLoss(torch.autograd.Function):
    """Makes `loss.backward()` of the reference trainer (policy_trainer.py:82) work: the forward already ran the
    fused forward+backward; backward hands the staged gradients to autograd (scaled by the incoming grad)."""

    @staticmethod
    def forward(ctx, policy, obs_features, actions, t, pair_i, pair_j, enc_noise, fm_noise, enc_latents, *params):
        pw, gw, gsd = policy._train_handles(stage=True)
        frozen = policy.frozen_prefixes()
        staged = [None if k.startswith(frozen) else gsd[k] for k, _ in policy.hot_path_named_parameters()]
        need_dobs = obs_features.requires_grad
        ctx.dlat = None
        if enc_latents is not None:     # an external action encoder: its output goes in, d loss / d output comes back
            losses, dobs, dlat = ops.compute_loss(pw, policy._loss_cfg, obs_features.detach(), actions, t, grads=gw, pair_i=pair_i,
                                                  pair_j=pair_j, need_dobs=need_dobs, enc_noise=enc_noise, fm_noise=fm_noise,
                                                  enc_latents=enc_latents.detach())
            ctx.dlat = dlat if enc_latents.requires_grad else None
        else:
            losses, dobs = ops.compute_loss(pw, policy._loss_cfg, obs_features.detach(), actions, t, grads=gw, pair_i=pair_i, pair_j=pair_j,
                                            need_dobs=need_dobs, enc_noise=enc_noise, fm_noise=fm_noise)
        ctx.staged, ctx.dobs, ctx.buf = staged, dobs, policy._cache["stage_buf"]
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, gout, _):
        ctx.buf.mul_(gout)      # ONE pass over the flat staging buffer (the incoming grad is a device scalar: no host check)
        dobs = ctx.dobs * gout if ctx.dobs is not None else None
        dlat = ctx.dlat * gout if ctx.dlat is not None else None
        return (None, dobs, None, None, None, None, None, None, dlat) + tuple(ctx.staged)


class _FusedLossExternalDecoder(torch.autograd.Function):
    """The fused step around an action decoder that is a torch module (vita_loss_config.external_decoder): forward = the
    forward call, returning the partial loss (every term but the decoder's) and the latents [pred; z1] as a differentiable
    output; backward = the resume_backward call, seeded with the incoming d / d latents.  The backward pass is linear in its
    seeds, so it runs on d_latents / gout with unit loss seeds and the staged gradients are scaled by gout afterwards
    (gout = 0 would lose the latents' part; autograd passes 1 for loss.backward())."""

    @staticmethod
    def forward(ctx, policy, obs_features, actions, t, pair_i, pair_j, enc_noise, fm_noise, enc_latents, *params):
        pw, _, _ = policy._train_handles(stage=True)
        obs_d = obs_features.detach()
        lat_d = enc_latents.detach() if enc_latents is not None else None
        out = ops.compute_loss(pw, policy._loss_cfg, obs_d, actions, t, pair_i=pair_i, pair_j=pair_j, enc_noise=enc_noise, fm_noise=fm_noise,
                               enc_latents=lat_d)
        losses, latents = out[0], out[-1]
        pw._step_generation = getattr(pw, "_step_generation", 0) + 1      # the saved activations live in the handle's one workspace
        ctx.generation = pw._step_generation
        ctx.policy, ctx.args = policy, (obs_d, actions, t, pair_i, pair_j, enc_noise, fm_noise, lat_d)
        ctx.need_dobs = obs_features.requires_grad
        ctx.need_dlat = enc_latents is not None and enc_latents.requires_grad
        ctx.losses = losses
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses, latents

    @staticmethod
    def backward(ctx, gout, _, glat):
        policy = ctx.policy
        obs_d, actions, t, pair_i, pair_j, enc_noise, fm_noise, lat_d = ctx.args
        pw, gw, gsd = policy._train_handles(stage=True)
        if getattr(pw, "_step_generation", 0) != ctx.generation:
            raise L.VitaHipError("compute_loss with an external action decoder: backward() must run before the next compute_loss call of "
                                 "this policy (the forward's saved activations live in one workspace); accumulate gradients by calling "
                                 "backward() per micro-batch")
        if gout is None:
            gout = torch.zeros((), dtype=torch.float32, device=obs_d.device)
        scale = torch.where(gout != 0, gout, torch.ones_like(gout))
        B = obs_d.shape[0]
        dl = glat / scale if glat is not None else torch.zeros((2 * B, policy.latent_dim), dtype=torch.float32, device=obs_d.device)
        out = ops.compute_loss(pw, policy._loss_cfg, obs_d, actions, t, grads=gw, pair_i=pair_i, pair_j=pair_j, need_dobs=ctx.need_dobs,
                               losses=ctx.losses, enc_noise=enc_noise, fm_noise=fm_noise, enc_latents=lat_d, dec_dlatents=dl)
        policy._cache["stage_buf"].mul_(scale)
        frozen = policy.frozen_prefixes()
        staged = [None if k.startswith(frozen) else gsd[k] for k, _ in policy.hot_path_named_parameters()]
        dobs = out[1] * scale if out[1] is not None else None
        dlat = out[2] * scale if (ctx.need_dlat and len(out) > 2 and out[2] is not None) else None
        return (None, dobs, None, None, None, None, None, None, dlat) + tuple(staged)


class _FusedMeanFlowLoss(torch.autograd.Function):
    """_FusedLoss for flow_matcher.name = 'mean': vita_compute_loss_meanflow_f32 (forward + backward in one call)."""

    @staticmethod
    def forward(ctx, policy, obs_features, actions, t, r, enc_noise, *params):
        pw, gw, gsd = policy._train_handles(stage=True)
        ctx.dobs = None
        if obs_features.requires_grad:      # a trainable observer upstream (vita_policy.py:113-114)
            losses, ctx.dobs = ops.compute_loss_meanflow(pw, policy._mf_cfg, obs_features.detach(), actions, t, r, grads=gw,
                                                         enc_noise=enc_noise, need_dobs=True)
        else:
            losses = ops.compute_loss_meanflow(pw, policy._mf_cfg, obs_features.detach(), actions, t, r, grads=gw, enc_noise=enc_noise)
        frozen = policy.frozen_prefixes()     # frozen modules keep .grad = None, as in the reference (and as _FusedLoss does)
        ctx.staged = [None if k.startswith(frozen) else gsd[k] for k, _ in policy.hot_path_named_parameters()]
        ctx.buf = policy._cache["stage_buf"]
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, gout, _):
        ctx.buf.mul_(gout)
        dobs = ctx.dobs * gout if ctx.dobs is not None else None
        return (None, dobs, None, None, None, None) + tuple(ctx.staged)


class VitaPolicy(nn.Module):
    """flare/policies/vita/vita_policy.py:36-283 + base_policy.py:15-89."""

    # flat-buffer segment order = the order gradients become final in the backward pass (vita_loss_step.hip)
    BUCKET_PREFIXES = (("action_decoder.",), ("flow_net.",), ("action_encoder.", "obs_encoder."))

    def __init__(self, config, stats, observer=None, action_encoder: Optional[nn.Module] = None,
                 action_decoder: Optional[nn.Module] = None):
        """observer: None (precomputed camera latents, FeatureObserver), "resnet18" (vita_amd.observers.ResNetObserver built from
        the config, vita_policy.py:43-50), or an nn.Module mapping the batch to (B, obs_dim) features.
        action_encoder: an nn.Module mapping normalised actions (B, T, A) to latents (B, latent_dim) that replaces the
        configured one (for instance the reference's own module of a non-default encoder type).  It runs under torch autograd
        and meets the fused path at the latents (VITA_ENC_EXTERNAL, vita_amd/encoders.py); `encoder_type: transformer` builds
        vita_amd.encoders.TransformerActionEncoder this way.  Under `use_variational` the module returns the (B, 2 * latent_dim)
        posterior parameters [mean | logvar]; compute_loss samples and adds the KL term (models/dists.py:4-35, vita_policy.py:146-150).
        action_decoder: an nn.Module mapping latents (B, latent_dim) to normalised actions (B, T, A) that replaces the configured
        one; the fused step then runs as a forward call and a backward call around it (vita_loss_config.external_decoder,
        vita_amd/decoders.py).  `use_variational` with `decoder_type: cnn` builds vita_amd.decoders.CNNActionDecoder this way."""
        super().__init__()
        self.config = config
        self.stats = stats
        pol, task = config.policy, config.task
        # base_policy.py:22-30
        self.pred_horizon = pol.pred_horizon
        self.action_horizon = pol.action_horizon
        self.obs_horizon = pol.obs_horizon
        self.action_dim = task.action_dim
        self.normalize_inputs = Normalize(list(task.image_keys) + [task.state_key], stats)
        self.normalize_targets = Normalize([task.action_key], stats)
        self.unnormalize_outputs = Unnormalize([task.action_key], stats)
        self._action_queue = None
        self._last_solve = None
        self.check_status = True       # select_action raises when the in-launch sampler timed out (check_sampler_status)
        # vita_policy.py:41-74
        self.num_sampling_steps = pol.flow_matcher.num_sampling_steps
        if isinstance(observer, str):
            if observer != "resnet18":
                raise ValueError(f"observer {observer!r}: 'resnet18', an nn.Module or None")
            from .observers import ResNetObserver
            observer = ResNetObserver(state_key=task.state_key, image_keys=task.image_keys, resize_shape=config.resize_shape,
                                      crop_shape=config.crop_shape, state_dim=task.state_dim, tokenize=False)    # :43-50
        self.observer = observer if observer is not None else FeatureObserver(task.state_key)
        self.obs_dim = len(task.image_keys) * 512 + task.state_dim
        self.FM = get_flow_matcher(**dict(pol.flow_matcher))
        ae = pol.action_ae
        net = ae.net
        self.use_action_vae = ae.use_variational
        self.latent_dim = net.latent_dim
        self.freeze_action_encoder = ae.freeze_encoder
        self.freeze_action_decoder = ae.freeze_decoder
        self.flow_action_recon_weight = ae.flow_recon_weight
        self.enc_action_recon_weight = ae.enc_recon_weight
        self.action_kl_weight = ae.kl_weight
        if ae.recon_loss_type not in ("l1", "l2"):
            raise ValueError(f"Unsupported recon_loss_type: {ae.recon_loss_type}. Use 'l1' or 'l2'.")   # :71
        self.enc_contrastive_weight = pol.vita.enc_contrastive_weight
        self.flow_contrastive_weight = pol.vita.flow_contrastive_weight
        self.obs_encoder = nn.Linear(self.obs_dim, self.latent_dim)                                      # :77
        self.external_encoder = action_encoder is not None or net.encoder_type == "transformer"
        if action_encoder is not None:
            self.action_encoder = action_encoder
        elif self.external_encoder:                                                                      # action_ae.py:354-364
            from .encoders import TransformerActionEncoder
            self.action_encoder = TransformerActionEncoder(enc_hidden_dim=net.enc_hidden_dim, latent_dim=self.latent_dim,
                                                           num_heads=net.num_heads, pred_horizon=self.pred_horizon,
                                                           action_dim=self.action_dim, num_layers=net.num_layers,
                                                           mlp_ratio=net.get("mlp_ratio", 4.0), dropout=net.get("dropout", 0.0),
                                                           variational=bool(self.use_action_vae))             # action_vae.py:408-418
        else:
            self.action_encoder = make_action_encoder(net.encoder_type, self.pred_horizon, self.action_dim, self.latent_dim,
                                                      net.enc_hidden_dim, net.num_layers, bool(self.use_action_vae))   # :101-110
        cnn_decoder = bool(self.use_action_vae) and net.decoder_type == "cnn"                            # action_vae.py:422-429
        self.external_decoder = action_decoder is not None or cnn_decoder
        if action_decoder is not None:
            self.action_decoder = action_decoder
        elif cnn_decoder:
            from .decoders import CNNActionDecoder
            self.action_decoder = CNNActionDecoder(pred_horizon=self.pred_horizon, action_dim=self.action_dim, latent_dim=self.latent_dim,
                                                   hidden_dim=net.get("dec_hidden_dim", 512), num_layers=net.get("num_layers", 4))
        else:
            # the variational factory takes every other decoder_type as 'simple' (action_vae.py:430-438); the plain one raises (action_ae.py:385-386)
            self.action_decoder = make_action_decoder("simple" if self.use_action_vae else net.decoder_type, self.pred_horizon,
                                                      self.action_dim, self.latent_dim, net.dec_hidden_dim, net.num_layers)
        fn = pol.flow_net
        if fn.name not in FLOW_NET_NAMES + (MEAN_FLOW_NET_NAME,):
            raise ValueError(f"Unsupported flow_net: {fn.name}")                                         # :89
        if float(fn.get("dropout", 0.0)) != 0.0:
            raise NotImplementedError("flow_net.dropout > 0 is not implemented (vita.yaml:32 uses 0.0)")
        # alternate matchers (mean / improved_mean / consistency): sampler only, see flow.py
        self.inference_only = bool(getattr(self.FM, "inference_only", False))
        self.mean_flow = fn.name == MEAN_FLOW_NET_NAME
        self._mf_cfg = None
        if self.mean_flow != isinstance(self.FM, MeanFlowMatcher):
            raise ValueError("flow_net 'simple_mean_flow_net' goes with the 'mean' / 'improved_mean' flow matchers, and only with them")
        if self.mean_flow and (self.external_encoder or self.external_decoder):
            raise NotImplementedError("an external action encoder / decoder (transformer, cnn) is built for the default (torchcfm) "
                                      "matchers' fused step only")
        if self.mean_flow:
            self.flow_net = make_mean_flow_net(self.latent_dim, fn.hidden_dim, fn.num_layers, fn.mlp_ratio)
        else:
            self.flow_net = make_flow_net(self.latent_dim, fn.hidden_dim, fn.num_layers, fn.mlp_ratio)  # :90-97
        # GEMM operand type: "f32" (default: the reference's strict-fp32 numerics), "bf16" (bf16 MFMA operands, fp32
        # accumulate: inference always, training for the GEMMs large enough to gain, vita_loss_config.gemm_bf16 = 1) or
        # "bf16_all" (training: every GEMM).  Stated tolerances in tests/test_gpu_bf16*.py.  Not a reference key.
        self.gemm_dtype = str(pol.get("gemm_dtype", "f32"))
        if self.gemm_dtype not in ("f32", "bf16", "bf16_all"):
            raise ValueError(f"policy.gemm_dtype must be 'f32', 'bf16' or 'bf16_all', got {self.gemm_dtype!r}")
        if self.external_decoder and self.gemm_dtype != "f32":
            raise NotImplementedError("an external action decoder goes with policy.gemm_dtype 'f32'")
        self._loss_cfg = ops.loss_config(num_steps=self.num_sampling_steps, recon=ae.recon_loss_type,
                                         decode_flow_latents=bool(pol.vita.decode_flow_latents),
                                         consistency_weight=float(pol.vita.consistency_weight),
                                         enc_recon_weight=float(self.enc_action_recon_weight),
                                         flow_recon_weight=float(self.flow_action_recon_weight),
                                         enc_contrastive_weight=float(self.enc_contrastive_weight),
                                         flow_contrastive_weight=float(self.flow_contrastive_weight),
                                         freeze_encoder=bool(self.freeze_action_encoder),
                                         freeze_decoder=bool(self.freeze_action_decoder),
                                         kl_weight=float(self.action_kl_weight) if self.use_action_vae else 0.0,
                                         gemm_bf16={"f32": 0, "bf16": 1, "bf16_all": 2}[self.gemm_dtype],
                                         fm_kind=int(self.FM.kind), fm_sigma=float(self.FM.sigma))
        if self.mean_flow and not self.inference_only:      # MeanFlow training: the fused MeanFlow step's branches
            self._mf_cfg = ops.meanflow_loss_config(num_steps=self.num_sampling_steps, recon=ae.recon_loss_type,
                                                    decode_flow_latents=bool(pol.vita.decode_flow_latents),
                                                    consistency_weight=float(pol.vita.consistency_weight),
                                                    enc_recon_weight=float(self.enc_action_recon_weight),
                                                    flow_recon_weight=float(self.flow_action_recon_weight),
                                                    adaptive_loss_gamma=float(self.FM.adaptive_loss_gamma), use_imf=bool(self.FM.use_imf),
                                                    norm_p=float(self.FM.norm_p), norm_eps=float(self.FM.norm_eps),
                                                    aux_v_loss_weight=float(self.FM.aux_v_loss_weight),
                                                    dispersive_loss_weight=float(self.FM.dispersive_loss_weight),
                                                    dispersive_loss_tau=float(self.FM.dispersive_loss_tau),
                                                    freeze_encoder=bool(self.freeze_action_encoder),
                                                    freeze_decoder=bool(self.freeze_action_decoder),
                                                    enc_contrastive_weight=float(self.enc_contrastive_weight),
                                                    flow_contrastive_weight=float(self.flow_contrastive_weight),
                                                    kl_weight=float(self.action_kl_weight) if self.use_action_vae else 0.0)
        self._flat = self._flat_grad = None
        self._cache = {}
        self._weights_epoch = 0
        self.bucket_ranges = None
        self.reset()

    def mark_weights_updated(self):
        """Tell the policy that parameter VALUES changed behind torch's back (the fused optimizer kernel and checkpoint loads
        write through raw pointers, which bumps no tensor version): derived copies (the packed bf16 weights) are rebuilt."""
        self._weights_epoch += 1

    # -- parameter plumbing --------------------------------------------------------------------------
    def hot_path_named_parameters(self):
        """(name, parameter) of the trainable hot-path tensors in flat-buffer order: decoder, flow net (time path and modulators,
        then the layers' mlp weights layer by layer, then the outer projections and the final norm: each gradient-ready event of
        reduce_plan covers ONE contiguous range and is one all-reduce), encoders."""
        named = dict(self.named_parameters())
        out = []
        for prefixes in self.BUCKET_PREFIXES:
            group = []
            for k, p in named.items():
                if (self.external_encoder and k.startswith("action_encoder.")) or (self.external_decoder and k.startswith("action_decoder.")):
                    continue        # a torch module beside the fused path: its gradients come through autograd
                if k.startswith(prefixes) and p.requires_grad:
                    group.append((k, p))
            group.sort(key=lambda kp: self._flow_event(kp[0]))     # stable: module order within each class
            out += group
        return out

    def observer_named_parameters(self):
        """(name, parameter) of the observer's trainable tensors (the ResNet18 backbone's convolutions: the reference trains them
        in the same AdamW group, vita_policy.py:222-229).  They follow the hot-path parameters in the flat buffers."""
        obs = getattr(self, "observer", None)
        if obs is None:
            return []
        return [(f"observer.{k}", p) for k, p in obs.named_parameters() if p.requires_grad]

    def flat_named_parameters(self):
        """Everything the native trainer's flat buffers hold, in buffer order: the hot path, then the observer."""
        return self.hot_path_named_parameters() + self.observer_named_parameters()

    def _flow_event(self, key: str) -> int:
        """Gradient-ready event of a flow_net.* tensor of the default flow net (vita_loss_config.grad_ready_layers): 3 + l for
        layers.l.mlp.*, 3 + L for the outer projections and the final norm, 1 for the time path (time_embed, time_modulator);
        1 for everything else and for the MeanFlow net (its step records events 0..2)."""
        parts = key.split(".")
        if parts[0] != "flow_net":
            return 1
        if self.mean_flow:      # vita_meanflow_loss_config.grad_ready_layers: hidden_layers.l -> 3 + l, embeddings -> 1, input layer / heads -> 3 + L
            if parts[1] == "hidden_layers":
                return 3 + int(parts[2])
            if parts[1] in ("t_embed", "h_embed"):
                return 1
            return 3 + int(self.config.policy.flow_net.num_layers)
        if len(parts) > 3 and parts[1] == "layers" and parts[3] == "mlp":
            return 3 + int(parts[2])
        if parts[1] in ("input_proj", "norm", "out_proj"):
            return 3 + int(self.config.policy.flow_net.num_layers)
        return 1

    def frozen_prefixes(self) -> tuple:
        """Modules that receive no gradient under the freeze flags (vita_policy.py:119,153,174,183): the reference leaves
        their .grad at None, so its optimizer skips them."""
        out = []
        if self.freeze_action_encoder:
            out.append("action_encoder.")
        if self.freeze_action_decoder:
            out.append("action_decoder.")
        return tuple(out) if out else ("\0",)

    def trainable_ranges(self):
        """Element ranges of the flat buffers the optimizer updates (everything but the frozen modules), merged."""
        frozen, off, out = self.frozen_prefixes(), 0, []
        for k, p in self.hot_path_named_parameters():
            n = p.numel()
            if not k.startswith(frozen):
                if out and out[-1][1] == off:
                    out[-1] = (out[-1][0], off + n)
                else:
                    out.append((off, off + n))
            off += n
        n_obs = sum(p.numel() for _, p in self.observer_named_parameters())
        if n_obs:
            if out and out[-1][1] == off:
                out[-1] = (out[-1][0], off + n_obs)
            else:
                out.append((off, off + n_obs))
        return out

    def reduce_plan(self):
        """[(lo, hi, event)] over the flat gradient buffer, in the order the events fire inside the fused backward
        (vita_loss_config.grad_ready_layers): the decoder (event 0); each layer's mlp.fc1 / mlp.fc2 gradients (event 3 + l, their
        stacked weight-gradient GEMMs run layer by layer); the encoders (event 2); the time path and the modulators (event 1); the
        outer projections and the final norm (event 3 + L) — most gradient bytes per unit of compute first, 2 MB last.  The
        MeanFlow step (round 3) records 0, 3 .. 3+L-1 (hidden layers), 1 (time / interval embeddings), 3+L (input layer, heads), 2."""
        plan, off = [], 0
        named = self.hot_path_named_parameters()
        for k, p in named:
            n = p.numel()
            if k.startswith("action_decoder."):
                ev = 0
            elif k.startswith("flow_net."):
                ev = self._flow_event(k)
            else:
                ev = 2
            if plan and plan[-1][2] == ev and plan[-1][1] == off:
                plan[-1] = (plan[-1][0], off + n, ev)
            else:
                plan.append((off, off + n, ev))
            off += n
        # 4 + L events fire 0, 3 .. 3+L-1, 2, 1, 3+L; 3 events only: 0, 1, 2
        last = max(ev for _, _, ev in plan)
        if self.mean_flow:      # fires 0, 3 .. 3+L-1, 1, 3+L, 2 (the encoders last)
            order = {0: -1, 1: last - 0.5, 2: last + 1}
        else:
            order = {0: -1, 2: last - 0.6, 1: last - 0.3} if last >= 3 else {}
        plan.sort(key=lambda seg: (order.get(seg[2], seg[2]), seg[0]))
        n_obs = sum(p.numel() for _, p in self.observer_named_parameters())
        if n_obs:       # the observer's gradients are final last: torch's backward of the backbone runs behind the fused step,
                        # which hands it d loss / d features; its event (index last + 1) is recorded by loss_and_grads
            plan.append((off, off + n_obs, last + 1))
        return plan

    def flatten_parameters(self):
        """Re-home the hot-path parameters (and their .grad) as views of two flat fp32 device buffers, so the data-parallel
        trainer all-reduces contiguous buckets and the optimizer is one fused kernel.  Call after .to(device)."""
        named = self.hot_path_named_parameters()
        dev = named[0][1].device
        if dev.type != "cuda":
            raise L.VitaHipError("flatten_parameters: move the policy to the GPU first (policy.to('cuda:0'))")
        obs_named = self.observer_named_parameters()
        n_hot = sum(p.numel() for _, p in named)
        n = n_hot + sum(p.numel() for _, p in obs_named)
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off, ranges, bucket = 0, [], 0
        starts = [0]
        for k, p in named:
            while not k.startswith(self.BUCKET_PREFIXES[bucket]):
                bucket += 1
                starts.append(off)
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        starts.append(off)
        while len(starts) < len(self.BUCKET_PREFIXES) + 1:
            starts.append(off)
        ranges = [(starts[i], starts[i + 1]) for i in range(len(self.BUCKET_PREFIXES))]
        for _, p in obs_named:      # the observer's trainable tensors behind the hot path: same buffers, same fused optimizer
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.observer_range = (n_hot, n) if n > n_hot else None
        self._flat, self._flat_grad, self.bucket_ranges = flat, flat_grad, ranges
        self._cache = {}
        return flat, flat_grad

    def _key(self):
        w = self.obs_encoder.weight
        last = self.flow_net.v_output_layer if self.mean_flow else self.flow_net.out_proj
        return (w.data_ptr(), w.device, last.bias.data_ptr())

    def _infer_handles(self):
        c = self._cache.get("infer")
        if c is None or c[0] != self._key():
            sd = {k: p.data for k, p in self.named_parameters()}
            dev = self.obs_encoder.weight.device
            if self.mean_flow:
                c = (self._key(), ops.MeanFlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev), None)
            else:
                c = (self._key(), ops.FlowNetHandle(sd, dev), None if self.external_decoder else ops.DecoderHandle(sd, dev),
                     ops.PolicyHandle(sd, dev, self.pred_horizon, **self._external_kw()))
            self._cache["infer"] = c
        return c[1], c[2], c[3]

    def _external_kw(self) -> dict:
        kw = {}
        if self.external_encoder:
            kw["external_encoder_action_dim"] = self.action_dim
        if self.external_decoder:
            kw["external_decoder_action_dim"] = self.action_dim
        return kw

    def _train_handles(self, stage: bool = False):
        """(weights, gradient descriptor, {name: gradient tensor}).  stage=False: the gradient tensors ARE the parameters'
        .grad (views of the flat gradient buffer after flatten_parameters) — the native trainer's path.  stage=True: separate
        staging tensors, for the autograd bridge (autograd then accumulates them into .grad itself)."""
        slot = "train_stage" if stage else "train"
        c = self._cache.get(slot)
        if c is None or c[0] != self._key():
            named = self.hot_path_named_parameters()
            sd = {k: p.data for k, p in named}
            if stage:
                n = sum(p.numel() for _, p in named)
                buf = torch.zeros(n, dtype=torch.float32, device=named[0][1].device)
                self._cache["stage_buf"] = buf
                gsd, off = {}, 0
                for k, p in named:
                    gsd[k] = buf[off:off + p.numel()].view_as(p)
                    off += p.numel()
            elif self._flat_grad is not None:
                gsd = {k: p.grad for k, p in named}
            else:
                gsd = {k: torch.zeros_like(p.data) for k, p in named}
            dev = self.obs_encoder.weight.device
            handle = ops.MeanFlowPolicyHandle if self.mean_flow else ops.PolicyHandle
            kw = self._external_kw()
            c = (self._key(), handle(sd, dev, self.pred_horizon, strict=True, **kw), handle(gsd, dev, self.pred_horizon, strict=True, **kw), gsd)
            self._cache[slot] = c
        return c[1], c[2], c[3]

    # -- base_policy.py API ----------------------------------------------------------------------------
    def get_action_indices(self):
        return list(range(self.pred_horizon))

    def get_observation_indices(self):
        return list(range(self.obs_horizon))

    def reset(self):
        self._action_queue = deque([], maxlen=self.action_horizon)

    def forward(self, batch):
        """base_policy.py:38-43."""
        batch = self.normalize_inputs(batch)
        batch = self.normalize_targets(batch)
        return self.compute_loss(batch)

    # -- the hot path -----------------------------------------------------------------------------------
    def _draw(self, batch_size, dev, t=None, pair_i=None, pair_j=None, obs_features=None, actions=None, enc_noise=None, fm_noise=None,
              enc_latents=None):
        """The host-side random draws of one compute_loss call, in the reference's order: the variational posterior's
        eps (torch.randn_like on the device, dists.py:20; training mode only, vita_policy.py:121), then t ~ U(0,1) and the
        OT pairing (flow_matchers.py:34 -> torchcfm), then the matcher's eps = randn_like(x0) when its xt uses it (sigma > 0,
        target, schrodinger).  Returns (t, pair_i, pair_j, enc_noise, fm_noise)."""
        if self.use_action_vae and self.training and enc_noise is None:
            enc_noise = torch.randn(batch_size, self.latent_dim, device=dev)
        if not (self.use_action_vae and self.training):
            enc_noise = None        # deterministic posterior: z = mean, KL = 0
        if t is None:
            t = torch.rand(batch_size).to(dev)
        if self.FM.pairing == "exact" and pair_i is None and pair_j is None:
            # the plan needs the latents: one encoder pass (forward only) — the reference pays the same sync
            _, _, pw = self._infer_handles()
            z0 = ops.linear(obs_features, self.obs_encoder.weight.data, self.obs_encoder.bias.data)
            z1 = enc_latents.detach() if enc_latents is not None else ops.action_encoder_forward(pw, actions)
            if self.use_action_vae and enc_latents is None:     # (B, 2D) rows [mean | logvar]; the sample feeds only the host-side OT plan here
                mean, logvar = z1[:, :self.latent_dim], z1[:, self.latent_dim:]
                z1 = mean if enc_noise is None else mean + torch.exp(0.5 * logvar) * enc_noise
            from .flow import exact_ot_pairing
            pair_i, pair_j = exact_ot_pairing(z0, z1)
        if self.FM.needs_noise and fm_noise is None:
            fm_noise = torch.randn(batch_size, self.latent_dim, device=dev)
        if not self.FM.needs_noise:
            fm_noise = None
        return t, pair_i, pair_j, enc_noise, fm_noise

    def loss_and_grads(self, batch, *, t=None, pair_i=None, pair_j=None, grad_ready_events=None, losses=None, enc_noise=None,
                       fm_noise=None, observer_event=None):
        """Native training entry: normalised batch -> losses (device float[8], _lib.LOSS_NAMES order), with d loss / d parameter
        written into the parameters' .grad buffers.  No host synchronisation.  With a trainable observer (observer='resnet18':
        the reference trains the backbone in the same AdamW group, vita_policy.py:113-114, 222-229) the fused step also returns
        d loss / d features and torch's backward of the observer alone turns it into the backbone's gradients, written into the
        observer's range of the same flat buffer; `observer_event` (a torch.cuda.Event) is recorded behind it."""
        if self.inference_only:
            raise NotImplementedError(f"training with flow matcher '{self.config.policy.flow_matcher.name}' is not built "
                                      "(SURVEY.md 8(f) #4): only its sampler runs on the HIP path")
        if self.external_encoder or self.external_decoder:
            raise NotImplementedError("the native trainer (flat buffers, fused optimizer) has no external action encoder / decoder "
                                      "(transformer, cnn): train it through compute_loss(...).backward() with a torch optimizer (the drop-in path)")
        obs_params = [p for _, p in self.observer_named_parameters()]
        with torch.set_grad_enabled(bool(obs_params)):      # the observer's graph only; the hot path has its own backward
            obs_features = self.observer(batch)
        need_dobs = bool(obs_params) and obs_features.requires_grad
        actions = batch[self.config.task.action_key]
        pw, gw, gsd = self._train_handles()
        if self._flat_grad is None:
            for (k, p) in self.hot_path_named_parameters():
                p.grad = gsd[k]
        if self.mean_flow:      # (t, r) of MeanFlowMatcher.sample_t_r; `pair_i` carries an explicit r
            if t is None:
                t, r = self.FM.sample_t_r(obs_features.shape[0], obs_features.device)
            else:
                r = pair_i if pair_i is not None else t
            if self.use_action_vae and self.training and enc_noise is None:      # the posterior's draw (action_vae.py:227, dists.py:19-22)
                enc_noise = torch.randn(obs_features.shape[0], self.latent_dim, device=obs_features.device)
            elif not self.use_action_vae or not self.training:
                enc_noise = None
            out = ops.compute_loss_meanflow(pw, self._mf_cfg, obs_features.detach(), actions, t, r.to(torch.float32), grads=gw, losses=losses,
                                            enc_noise=enc_noise, grad_ready_events=grad_ready_events, need_dobs=need_dobs)
            dobs = None
            if need_dobs:
                out, dobs = out
            self._observer_backward(obs_features, dobs, obs_params, observer_event)
            return out
        t, pair_i, pair_j, enc_noise, fm_noise = self._draw(obs_features.shape[0], obs_features.device, t, pair_i, pair_j, obs_features.detach(),
                                                            actions, enc_noise, fm_noise)
        out, dobs = ops.compute_loss(pw, self._loss_cfg, obs_features.detach(), actions, t, grads=gw, pair_i=pair_i, pair_j=pair_j,
                                     grad_ready_events=grad_ready_events, losses=losses, enc_noise=enc_noise, fm_noise=fm_noise,
                                     need_dobs=need_dobs)
        self._observer_backward(obs_features, dobs, obs_params, observer_event)
        return out

    def _observer_backward(self, obs_features, dobs, obs_params, event):
        """d loss / d features -> the observer's parameter gradients (torch autograd over the backbone alone), overwriting what the
        buffers held: the native step's gradients are written, never accumulated."""
        if dobs is not None:
            if self._flat_grad is not None and getattr(self, "observer_range", None):
                lo, hi = self.observer_range
                self._flat_grad[lo:hi].zero_()          # autograd accumulates in place into the views
            else:
                for p in obs_params:
                    p.grad = None
            torch.autograd.backward(obs_features, dobs, inputs=obs_params)
        if event is not None:
            event.record(torch.cuda.current_stream(obs_features.device))

    def compute_loss(self, batch, *, t=None, pair_i=None, pair_j=None, enc_noise=None, fm_noise=None):
        """vita_policy.py:112-189.  Returns (loss, metrics) with the reference's metric keys; `loss` supports
        .backward() (drop-in for policy_trainer.py:79-82).  One device->host copy for all metrics (the reference
        does one .item() per metric)."""
        if self.inference_only:
            raise NotImplementedError(f"training with flow matcher '{self.config.policy.flow_matcher.name}' is not built "
                                      "(SURVEY.md 8(f) #4): only its sampler runs on the HIP path")
        obs_features = self.observer(batch)
        actions = batch[self.config.task.action_key]
        if self.mean_flow:
            return self._compute_loss_meanflow(obs_features, actions, t, pair_i, enc_noise)
        enc_latents, kl_rows = None, None
        if self.external_encoder:       # :118-123 — under no_grad when the encoder is frozen
            with torch.set_grad_enabled(torch.is_grad_enabled() and not self.freeze_action_encoder):
                enc_latents = self.action_encoder(actions)
                if self.use_action_vae:     # DiagonalGaussianDistribution(params, deterministic=not training), dists.py:4-35
                    mean, logvar = torch.chunk(enc_latents, 2, dim=1)
                    if self.training:
                        if enc_noise is None:
                            enc_noise = torch.randn_like(mean)
                        enc_latents = mean + torch.exp(0.5 * logvar) * enc_noise
                        kl_rows = 0.5 * torch.sum(torch.pow(mean, 2) + torch.exp(logvar) - 1.0 - logvar, dim=1)
                    else:
                        enc_latents, kl_rows = mean, torch.zeros(mean.shape[0], device=mean.device)
                    enc_latents = enc_latents.contiguous()
        t, pair_i, pair_j, enc_noise, fm_noise = self._draw(obs_features.shape[0], obs_features.device, t, pair_i, pair_j, obs_features,
                                                            actions, enc_noise, fm_noise, enc_latents)
        if self.external_encoder:
            enc_noise = None        # the posterior's draw was used above; the fused call sees plain latents
        with_grad = torch.is_grad_enabled() and any(p.requires_grad for _, p in self.hot_path_named_parameters())
        latents = None
        if with_grad:
            params = [p for _, p in self.hot_path_named_parameters()]
            if self.external_decoder:
                loss, losses, latents = _FusedLossExternalDecoder.apply(self, obs_features, actions, t, pair_i, pair_j, enc_noise, fm_noise,
                                                                        enc_latents, *params)
            else:
                loss, losses = _FusedLoss.apply(self, obs_features, actions, t, pair_i, pair_j, enc_noise, fm_noise, enc_latents, *params)
        else:
            _, _, pw = self._infer_handles()
            out = ops.compute_loss(pw, self._loss_cfg, obs_features, actions, t, pair_i=pair_i, pair_j=pair_j, enc_noise=enc_noise,
                                   fm_noise=fm_noise, enc_latents=enc_latents.detach() if enc_latents is not None else None)
            losses = out[0]
            loss = losses[0]
            if self.external_decoder:
                latents = out[-1]
        if self.external_decoder or kl_rows is not None:    # the terms of modules that run under torch autograd beside the fused call
            loss, losses = self._external_terms(loss, losses, latents, actions, kl_rows)
        vals = losses.detach().cpu().tolist()
        metrics = {"loss": vals[1], "flow_loss": vals[1]}   # FM.compute_loss's own 'loss' entry is the flow loss (flow_matchers.py:39)
        cfgv = self.config.policy.vita
        if self.enc_contrastive_weight > 0:
            metrics["enc_contrastive_loss"] = vals[5]
        fe, fd = self.freeze_action_encoder, self.freeze_action_decoder
        if not fe and self.use_action_vae and self.action_kl_weight > 0 and not (fe and fd):      # :142-150
            metrics["action_kl_loss"] = vals[7]
        if cfgv.decode_flow_latents and not fe and not fd:
            if cfgv.consistency_weight > 0:
                metrics["consistency_loss"] = vals[2]
            if self.flow_contrastive_weight > 0:
                metrics["flow_contrastive_loss"] = vals[6]
            if self.flow_action_recon_weight > 0:
                metrics["flow_action_recon_loss"] = vals[3]
        if self.enc_action_recon_weight > 0 and not fd:
            metrics["enc_action_recon_loss"] = vals[4]
        return loss, metrics

    def _external_terms(self, loss, losses, latents, actions, kl_rows):
        """The loss terms of modules outside the fused call: an external variational encoder's KL (vita_policy.py:146-150) and an
        external decoder's reconstruction terms (:174-187) on the latents [pred; z1] the fused forward handed out.  Returns
        (loss + weighted terms, the losses vector with those entries filled in)."""
        B = actions.shape[0]
        recon = F.l1_loss if self.config.policy.action_ae.recon_loss_type == "l1" else F.mse_loss
        fe, fd = self.freeze_action_encoder, self.freeze_action_decoder
        terms = {}
        if not (fe and fd) and kl_rows is not None and not fe and self.action_kl_weight > 0:         # :142-150
            terms[L.LOSS_NAMES.index("action_kl_loss")] = (self.action_kl_weight, kl_rows.mean())
        if not (fe and fd) and self.external_decoder:                                                # :142-143
            if self.config.policy.vita.decode_flow_latents and not fe and not fd and self.flow_action_recon_weight > 0:
                terms[L.LOSS_NAMES.index("flow_action_recon_loss")] = (self.flow_action_recon_weight, recon(self.action_decoder(latents[:B]), actions))
            if self.enc_action_recon_weight > 0 and not fd:
                terms[L.LOSS_NAMES.index("enc_action_recon_loss")] = (self.enc_action_recon_weight, recon(self.action_decoder(latents[B:]), actions))
        if terms:
            losses = losses.clone()
            for i, (w, v) in terms.items():
                loss = loss + w * v
                losses[i] = v.detach()
            losses[0] = loss.detach()
        return loss, losses

    def _compute_loss_meanflow(self, obs_features, actions, t=None, r=None, enc_noise=None):
        """compute_loss for flow_matcher.name = 'mean' (mean_flow_matcher.py:109-199 inside vita_policy.py:112-189); metric keys
        as the reference emits them: FM's {'meanflow_loss', 'loss'} plus the policy's."""
        if t is None:
            t, r = self.FM.sample_t_r(obs_features.shape[0], obs_features.device)
        elif r is None:
            r = t
        if self.use_action_vae and self.training and enc_noise is None:      # the posterior's draw (action_vae.py:227, dists.py:19-22)
            enc_noise = torch.randn(obs_features.shape[0], self.latent_dim, device=obs_features.device)
        elif not self.use_action_vae or not self.training:
            enc_noise = None        # deterministic posterior: z = mean, KL = 0
        if torch.is_grad_enabled() and any(p.requires_grad for _, p in self.hot_path_named_parameters()):
            params = [p for _, p in self.hot_path_named_parameters()]
            loss, losses = _FusedMeanFlowLoss.apply(self, obs_features, actions, t, r, enc_noise, *params)
        else:
            pw, _, _ = self._train_handles()
            losses = ops.compute_loss_meanflow(pw, self._mf_cfg, obs_features, actions, t, r, enc_noise=enc_noise)
            loss = losses[0]
        vals = losses.detach().cpu().tolist()
        if self.FM.use_imf:     # mean_flow_matcher.py:158-163, 197
            metrics = {"imf_loss": vals[5], "aux_v_loss": vals[6], "imf_mse": vals[7], "aux_v_mse": vals[8], "loss": vals[1], "flow_loss": vals[1]}
        else:
            metrics = {"meanflow_loss": vals[1], "loss": vals[1], "flow_loss": vals[1]}
        if self._mf_cfg.dispersive_loss_weight > 0:     # :188-195; 'meanflow_loss' was read before the dispersive term was added
            metrics["dispersive_loss"] = vals[9]
            if not self.FM.use_imf and obs_features.shape[0] > 1:
                metrics["meanflow_loss"] = vals[10]
        cfgv = self.config.policy.vita
        fe, fd = self.freeze_action_encoder, self.freeze_action_decoder
        if self.enc_contrastive_weight > 0:
            metrics["enc_contrastive_loss"] = vals[11]
        if fe and fd:                                                                              # :142-143
            return loss, metrics
        if not fe and self.use_action_vae and self.action_kl_weight > 0:                           # :146-150
            metrics["action_kl_loss"] = vals[13]
        if cfgv.decode_flow_latents and not fe and not fd:
            if cfgv.consistency_weight > 0:
                metrics["consistency_loss"] = vals[2]
            if self.flow_contrastive_weight > 0:
                metrics["flow_contrastive_loss"] = vals[12]
            if self.flow_action_recon_weight > 0:
                metrics["flow_action_recon_loss"] = vals[3]
        if self.enc_action_recon_weight > 0 and not fd:
            metrics["enc_action_recon_loss"] = vals[4]
        return loss, metrics

    @torch.no_grad()
    def generate_actions(self, batch):
        """vita_policy.py:191-208: observer features -> obs_encoder -> K Euler steps -> decoder, one C-ABI call."""
        obs_features = self.observer(batch)
        fw, dw, _ = self._infer_handles()
        if self.inference_only or self.mean_flow or self.external_decoder:
            # mean / improved_mean / consistency, or a decoder that is a torch module: obs_encoder -> FM.sample -> decoder
            z0 = ops.linear(obs_features, self.obs_encoder.weight.data, self.obs_encoder.bias.data)
            z = self.FM.sample(fw, shape=tuple(z0.shape), device=z0.device, start=z0, num_steps=self.num_sampling_steps)
            if self.external_decoder:
                return self.action_decoder(z).reshape(-1, self.pred_horizon, self.action_dim)
            return ops.decoder_forward(dw, z).view(-1, self.pred_horizon, self.action_dim)
        if self.gemm_dtype != "f32":
            out = ops.generate_actions_bf16(self.obs_encoder.bias.data, fw, dw, self._bf16_weights(), obs_features,
                                            self.num_sampling_steps)
        else:
            out = ops.generate_actions(self.obs_encoder.weight.data, self.obs_encoder.bias.data, fw, dw, obs_features,
                                       self.num_sampling_steps)
        self._last_solve = (obs_features.shape[0], obs_features.shape[1], obs_features.device)
        return out.view(-1, self.pred_horizon, self.action_dim)

    def check_sampler_status(self):
        """Raises VitaHipError when the last generate_actions' in-launch sampler timed out (a bounded hand-off wait ran out:
        csrc/vita_ode_persistent.hip TeamSync, csrc/vita_ode_resident.hip GridSync).  Its outputs are NaN then (the library
        poisons them on the device); this is the host-side half.  SYNCHRONISES: select_action and validate call it where they
        wait for the device anyway."""
        last = getattr(self, "_last_solve", None)
        if last is None:
            return
        B, O, device = last
        fw, dw, _ = self._infer_handles()
        if self.gemm_dtype != "f32":
            st = ops.generate_actions_bf16_status(fw, dw, O, B, self.num_sampling_steps, device)
        else:
            st = ops.generate_actions_status(fw, dw, B, self.num_sampling_steps, device)
        if st != 0:
            raise L.VitaHipError(f"the in-launch Euler sampler timed out in phase {st - 1} (B={B}, K={self.num_sampling_steps}): another kernel "
                                 "held compute units its workgroups needed; the actions of that call are NaN.  Re-run, or select the "
                                 "per-launch path with vita_amd.ops.set_persistent_sampler(False)")

    def _bf16_weights(self):
        """bf16 copies of the GEMM weights, re-packed whenever a parameter has been written since the last pack."""
        named = self.hot_path_named_parameters()
        version = (self._key(), self._weights_epoch, tuple(p._version for _, p in named))
        c = self._cache.get("bf16")
        if c is None or c[0] != version:
            c = (version, ops.Bf16Weights({k: p.data for k, p in named}, self.obs_encoder.weight.device))
            self._cache["bf16"] = c
        return c[1]

    @torch.no_grad()
    def select_action(self, batch):
        """vita_policy.py:210-220."""
        if self.training:       # the reference calls eval() on every step; walking the module tree costs more host time than the solve's launch
            self.eval()
        keep = list(self.config.task.image_keys) + [self.config.task.state_key, getattr(self.observer, "feature_key", None)]
        batch = {k: v.unsqueeze(1) for k, v in batch.items() if k in keep}
        batch = self.normalize_inputs(batch)
        if len(self._action_queue) == 0:
            actions = self.generate_actions(batch)
            if self.check_status:       # the caller copies the action to the host next (utils/eval.py:99): this wait is that wait
                self.check_sampler_status()
            actions = actions[:, :self.action_horizon]
            actions = self.unnormalize_outputs({"action": actions})["action"]
            self._action_queue.extend(actions.transpose(0, 1))
        return self._action_queue.popleft()

    def validate(self, batch):
        """base_policy.py:60-89 — the reference's chunks_per_sec meter (the benchmark's metric definition)."""
        self.eval()
        batch = self.normalize_inputs(batch)
        ak = self.config.task.action_key
        batch_size = batch[ak].shape[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred_norm = self.generate_actions(batch)
        torch.cuda.synchronize()
        gen_time_ms = (time.perf_counter() - t0) * 1000 / batch_size
        self.check_sampler_status()
        pred = self.unnormalize_outputs({"action": pred_norm})["action"]
        gt = batch[ak]
        with torch.no_grad():
            loss, metrics = self.compute_loss(batch)
        metrics["action_mse"] = ops.mse_loss_fwd(pred, gt).item()
        metrics["loss"] = loss.item()
        total_actions = batch_size * self.action_horizon
        metrics["time_per_chunk_ms"] = gen_time_ms
        metrics["time_per_action_ms"] = gen_time_ms / total_actions
        metrics["chunks_per_sec"] = 1000 / gen_time_ms
        return metrics

    def visualize(self, batch, num_samples: int = 1):
        """Plotting is outside the hot path; the trainer treats the result as optional."""
        return {}

    # -- optimizer / schedule / EMA factories (vita_policy.py:222-237, base_policy.py:45-49) ---------------
    def get_optimizer(self):
        c = self.config
        return torch.optim.AdamW(params=self.parameters(), lr=c.optimizer_lr, betas=tuple(c.optimizer_betas), eps=c.optimizer_eps,
                                 weight_decay=c.optimizer_weight_decay)

    def get_scheduler(self, optimizer, num_training_steps: int):
        """diffusers==0.30.3 get_scheduler('cosine' | 'constant' | 'constant_with_warmup') restated (UNPINNED)."""
        name, warm = self.config.scheduler_name, self.config.scheduler_warmup_steps
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: lr_lambda(name, s, warm, num_training_steps))

    def get_ema(self):
        return EMAModel(self.parameters(), power=self.config.ema_power) if self.config.use_ema else None

    # -- checkpoints: model.safetensors as PyTorchModelHubMixin.save_pretrained writes it (policy_trainer.py:265) ----
    def save_pretrained(self, save_directory: str):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        save_file({k: v.detach().cpu().clone().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "model.safetensors"))

    def load_pretrained(self, directory: str, strict: bool = False):
        """checkpoints.py:121-153: model.safetensors, load_state_dict(strict=False)."""
        from safetensors.torch import load_file
        return self.load_state_dict(load_file(os.path.join(directory, "model.safetensors")), strict=strict)


def lr_lambda(name: str, step: int, num_warmup_steps: int, num_training_steps: int) -> float:
    if name == "constant":
        return 1.0
    if name == "constant_with_warmup":
        return min(1.0, float(step) / float(max(1, num_warmup_steps)))
    if name == "cosine":
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    raise ValueError(f"scheduler {name!r} is not implemented (cosine, constant, constant_with_warmup)")


def ema_decay(optimization_step: int, max_decay: float = 0.9999, min_decay: float = 0.0, inv_gamma: float = 1.0,
              power: float = 2.0 / 3.0, use_ema_warmup: bool = False) -> float:
    """diffusers==0.30.3 EMAModel.get_decay restated (UNPINNED).  The reference passes only `power`
    (base_policy.py:45-49), so use_ema_warmup is False and decay = min((1+s)/(10+s), 0.9999)."""
    step = max(0, optimization_step - 1)
    if step <= 0:
        return 0.0
    cur = 1 - (1 + step / inv_gamma) ** -power if use_ema_warmup else (1 + step) / (10 + step)
    return max(min(cur, max_decay), min_decay)


class EMAModel:
    """The slice of diffusers.training_utils.EMAModel the runner uses: step(parameters), copy_to, state_dict."""

    def __init__(self, parameters, power: float = 2.0 / 3.0, decay: float = 0.9999):
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.power, self.decay, self.optimization_step, self.cur_decay_value = power, decay, 0, 0.0

    @torch.no_grad()
    def step(self, parameters):
        self.optimization_step += 1
        d = ema_decay(self.optimization_step, max_decay=self.decay, power=self.power)
        self.cur_decay_value = d
        for s, p in zip(self.shadow_params, list(parameters)):
            if p.requires_grad:
                s.sub_((1 - d) * (s - p))
            else:
                s.copy_(p)

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, list(parameters)):
            p.data.copy_(s.to(p.device).data)

    def state_dict(self):
        return {"decay": self.decay, "power": self.power, "optimization_step": self.optimization_step,
                "shadow_params": self.shadow_params}

    def load_state_dict(self, sd):
        self.decay, self.power, self.optimization_step = sd["decay"], sd["power"], sd["optimization_step"]
        self.shadow_params = [p.clone() for p in sd["shadow_params"]]
