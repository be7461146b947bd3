"""An action decoder that runs as a torch module beside the fused step (vita_loss_config.external_decoder: the forward call
hands out [pred; z1], the module forms the reconstruction terms under torch autograd, the resume_backward call takes
d terms / d latents and unrolls the rest).
1. The hook itself: a torch re-statement of SimpleActionDecoder (action_ae.py:226-279) plugged in as `action_decoder=` must give
   the losses and every gradient of the policy whose decoder runs inside the fused call; freeze flags; together with an
   external encoder; gradient scaling by the incoming grad.
2. `use_variational` + `decoder_type: cnn` (vita_amd.decoders.CNNActionDecoder, pinned against the reference's module by
   tests/test_decoders_cpu.py): one training step against fp64 torch autograd of the same composite loss built from the
   oracle's pieces; generate_actions / select_action decode with the module."""
import pytest
import torch
import torch.nn as nn

from oracle import vita_oracle as O
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cfg(**ae):
    cfg = default_config("robomimic_square")
    cfg.policy.vita.latent_dim = 64
    cfg.policy.action_ae.net.merge({"latent_dim": 64, "enc_hidden_dim": 64, "dec_hidden_dim": 48, "encoder_type": "cnn", "num_layers": 2})
    cfg.policy.flow_net.merge({"hidden_dim": 64, "num_layers": 2})
    cfg.policy.flow_matcher.name = "conditional"
    for k, v in ae.items():
        setattr(cfg.policy.action_ae, k, v)
    return cfg


def _stats(cfg):
    s, a = cfg.task.state_dim, cfg.task.action_dim
    return {cfg.task.state_key: {"mean": [0.0] * s, "std": [1.0] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [1.0] * a}}


def _batch(cfg, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {cfg.task.state_key: torch.randn(B, 1, cfg.task.state_dim, generator=g).to(DEV),
            "observation.features": torch.relu(torch.randn(B, 1, 512, generator=g)).to(DEV),
            cfg.task.action_key: torch.randn(B, 16, cfg.task.action_dim, generator=g).to(DEV)}, torch.rand(B, generator=g).to(DEV)


class TorchSimpleDecoder(nn.Module):
    """SimpleActionDecoder in plain torch, parameters under the reference's names."""

    def __init__(self, D, HD, NL, T, A):
        super().__init__()
        self.T, self.A = T, A
        self.input_proj = nn.Linear(D, HD)
        self.layers = nn.ModuleList([nn.ModuleDict({"fc1": nn.Linear(HD, HD), "fc2": nn.Linear(HD, HD)}) for _ in range(NL)])
        self.output_proj = nn.Linear(HD, T * A)

    def forward(self, z):
        sd = {"action_decoder." + k: v for k, v in self.named_parameters()}
        return O.simple_action_decode(sd, z, self.T, self.A)


def _pair(cfg, **kw):
    """(native policy, the same policy with its decoder as an external torch module, that module)"""
    torch.manual_seed(0)
    native = VitaPolicy(cfg, _stats(cfg)).to(DEV)
    A = cfg.task.action_dim
    mod = TorchSimpleDecoder(64, 48, 2, 16, A).to(DEV)
    nsd = native.state_dict()
    mod.load_state_dict({k[len("action_decoder."):]: v for k, v in nsd.items() if k.startswith("action_decoder.")}, strict=True)
    ext = VitaPolicy(cfg, _stats(cfg), action_decoder=mod, **kw).to(DEV)
    ext.load_state_dict({k: v for k, v in nsd.items() if not k.startswith("action_decoder.")}, strict=False)
    assert ext.external_decoder and not any(k.startswith("action_decoder.") for k, _ in ext.hot_path_named_parameters())
    return native, ext, mod


def _compare(native, ext, batch, t, scale=None):
    native.train(); ext.train()
    native.zero_grad(set_to_none=True); ext.zero_grad(set_to_none=True)
    torch.manual_seed(5); ln, mn = native.compute_loss(batch, t=t)
    torch.manual_seed(5); le, me = ext.compute_loss(batch, t=t)
    if scale is None:
        ln.backward(); le.backward()
    else:
        (ln * scale).backward(); (le * scale).backward()
    assert set(mn) == set(me)
    for k in mn:
        assert abs(mn[k] - me[k]) <= 2e-5 * max(1.0, abs(mn[k])), (k, mn[k], me[k])
    assert abs(ln.item() - le.item()) <= 2e-5 * max(1.0, abs(ln.item()))
    gn = {k: p.grad for k, p in native.named_parameters() if p.grad is not None}
    ge = {k: p.grad for k, p in ext.named_parameters() if p.grad is not None}
    assert set(gn) == set(ge), set(gn) ^ set(ge)
    for k, g in gn.items():
        err = ((g - ge[k]).abs().max() / g.abs().max().clamp_min(1e-20)).item()
        assert err <= 5e-5, (k, err)
    return gn


def test_external_decoder_equals_the_fused_one():
    cfg = _cfg()
    native, ext, mod = _pair(cfg)
    batch, t = _batch(cfg, 24)
    gn = _compare(native, ext, batch, t)
    assert "action_decoder.output_proj.weight" in gn and gn["action_decoder.output_proj.weight"].abs().max().item() > 0
    # a second step on another batch size (the step workspace is re-made), gradient scaled by the incoming grad
    batch2, t2 = _batch(cfg, 40, seed=3)
    _compare(native, ext, batch2, t2, scale=0.25)
    # no-grad evaluation (validate's call): forward call + module under no_grad
    with torch.no_grad():
        ln, mn = native.compute_loss(batch, t=t)
        le, me = ext.compute_loss(batch, t=t)
    assert abs(ln.item() - le.item()) <= 2e-5 * abs(ln.item()) and set(mn) == set(me)
    # inference decodes with the module
    native.eval(); ext.eval()
    an, ae = native.generate_actions(native.normalize_inputs(dict(batch))), ext.generate_actions(ext.normalize_inputs(dict(batch)))
    assert an.shape == ae.shape and (an - ae).abs().max().item() <= 2e-5
    with pytest.raises(NotImplementedError):
        ext.loss_and_grads(batch, t=t)
    # two forwards before a backward: the first graph's saved activations are gone, and it says so instead of using the second's
    ext.train()
    l1, _ = ext.compute_loss(batch, t=t)
    l2, _ = ext.compute_loss(batch2, t=t2)
    from vita_amd._lib import VitaHipError
    with pytest.raises(VitaHipError):
        l1.backward()
    l2.backward()


@pytest.mark.parametrize("flags", [dict(freeze_decoder=True), dict(freeze_encoder=True), dict(flow_recon_weight=0.0),
                                   dict(enc_recon_weight=0.0), dict(recon_loss_type="l2")])
def test_external_decoder_branches(flags):
    cfg = _cfg(**flags)
    native, ext, mod = _pair(cfg)
    batch, t = _batch(cfg, 16, seed=7)
    _compare(native, ext, batch, t)


def test_external_decoder_with_the_variational_encoder_and_contrastive_terms():
    cfg = _cfg(use_variational=True, kl_weight=1e-3)
    cfg.policy.action_ae.net.decoder_type = "simple"
    cfg.policy.vita.enc_contrastive_weight = 0.1
    cfg.policy.vita.flow_contrastive_weight = 0.05
    native, ext, mod = _pair(cfg)
    batch, t = _batch(cfg, 16, seed=9)
    noise = torch.randn(16, 64, generator=torch.Generator().manual_seed(4)).to(DEV)
    native.train(); ext.train()
    ln, mn = native.compute_loss(batch, t=t, enc_noise=noise)
    le, me = ext.compute_loss(batch, t=t, enc_noise=noise)
    ln.backward(); le.backward()
    assert set(mn) == set(me) and "action_kl_loss" in mn and "enc_contrastive_loss" in mn
    for k in mn:
        assert abs(mn[k] - me[k]) <= 2e-5 * max(1.0, abs(mn[k])), (k, mn[k], me[k])
    ge = dict(ext.named_parameters())
    for k, p in native.named_parameters():
        if p.grad is None:
            continue
        err = ((p.grad - ge[k].grad).abs().max() / p.grad.abs().max().clamp_min(1e-20)).item()
        assert err <= 5e-5, (k, err)


def test_external_decoder_and_external_encoder_together():
    from vita_amd.encoders import TransformerActionEncoder
    cfg = _cfg()
    cfg.policy.action_ae.net.encoder_type = "transformer"
    torch.manual_seed(2)
    native = VitaPolicy(cfg, _stats(cfg)).to(DEV)           # transformer encoder (external), fused simple decoder
    assert isinstance(native.action_encoder, TransformerActionEncoder)
    A = cfg.task.action_dim
    mod = TorchSimpleDecoder(64, 48, 2, 16, A).to(DEV)
    nsd = native.state_dict()
    mod.load_state_dict({k[len("action_decoder."):]: v for k, v in nsd.items() if k.startswith("action_decoder.")}, strict=True)
    ext = VitaPolicy(cfg, _stats(cfg), action_decoder=mod).to(DEV)
    ext.load_state_dict({k: v for k, v in nsd.items() if not k.startswith("action_decoder.")}, strict=False)
    assert ext.external_decoder and ext.external_encoder
    batch, t = _batch(cfg, 12, seed=11)
    _compare(native, ext, batch, t)


def test_cnn_decoder_policy_trains_and_matches_autograd():
    from vita_amd.decoders import CNNActionDecoder
    cfg = _cfg(use_variational=True, kl_weight=1e-3)
    cfg.policy.action_ae.net.decoder_type = "cnn"
    torch.manual_seed(1)
    pol = VitaPolicy(cfg, _stats(cfg)).to(DEV)
    assert isinstance(pol.action_decoder, CNNActionDecoder) and pol.external_decoder
    with torch.no_grad():       # the reference zero-initialises the biases: make them count
        for n, p in pol.action_decoder.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    B = 16
    batch, t = _batch(cfg, B, seed=2)
    noise = torch.randn(B, 64, generator=torch.Generator().manual_seed(4)).to(DEV)
    pol.train()
    loss, metrics = pol.compute_loss(batch, t=t, enc_noise=noise)
    loss.backward()
    assert {"loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss", "action_kl_loss"} <= set(metrics)
    # the same composite loss in fp64 torch autograd: the oracle's pieces with its restatement of the CNN decoder
    sd64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in pol.state_dict().items() if v.dtype == torch.float32}
    obs = pol.observer(batch).double().cpu()
    actions = batch[cfg.task.action_key].double().cpu()
    HID = pol.action_decoder.hidden_dim
    ref_loss, ref_m = O.compute_loss(sd64, obs, actions, t.double().cpu(), 6, kl_weight=1e-3, enc_noise=noise.double().cpu(),
                                     decoder=lambda z: O.cnn_action_decode(sd64, z, 16, HID))
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    for k in ("flow_action_recon_loss", "enc_action_recon_loss", "consistency_loss", "action_kl_loss"):
        assert abs(metrics[k] - ref_m[k].item()) <= 2e-5 * max(1.0, abs(ref_m[k].item())), k
    for k, p in pol.named_parameters():
        if p.grad is None:
            continue
        ref = sd64[k].grad
        assert ref is not None, k
        err = ((p.grad.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
        assert err <= 1e-4, (k, err)
    assert pol.action_decoder.latent_proj.weight.grad is not None and pol.obs_encoder.weight.grad is not None
    # one optimizer step through the reference's factories moves the decoder too
    opt = pol.get_optimizer()
    before = pol.action_decoder.latent_proj.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, pol.action_decoder.latent_proj.weight)
    # inference: sampler on the HIP path, the module decodes; against the oracle's pieces
    pol.eval()
    acts = pol.generate_actions(pol.normalize_inputs(dict(batch)))
    sd64 = {k: v.detach().double().cpu() for k, v in pol.state_dict().items() if v.dtype == torch.float32}
    z = O.euler_sample(sd64, O.obs_encode(sd64, obs), 6)
    ref_a = O.cnn_action_decode(sd64, z, 16, HID)
    assert tuple(acts.shape) == (B, 16, cfg.task.action_dim) and (acts.double().cpu() - ref_a).abs().max().item() <= 2e-5
    pol.reset()
    a = pol.select_action({k: v[:, 0] for k, v in batch.items() if k != cfg.task.action_key})
    assert tuple(a.shape) == (B, cfg.task.action_dim)


def test_variational_transformer_encoder_with_cnn_decoder_matches_autograd():
    """action_ae.use_variational with encoder_type transformer and decoder_type cnn (action_vae.py:390-440): both modules run
    under torch autograd around the fused step; the KL term and the posterior's sample are formed in the policy
    (models/dists.py).  Against the same composite loss in fp64 torch autograd over the oracle's pieces."""
    from vita_amd.decoders import CNNActionDecoder
    from vita_amd.encoders import TransformerActionEncoder
    cfg = _cfg(use_variational=True, kl_weight=1e-2)
    cfg.policy.action_ae.net.merge({"encoder_type": "transformer", "decoder_type": "cnn"})
    torch.manual_seed(3)
    pol = VitaPolicy(cfg, _stats(cfg)).to(DEV)
    assert isinstance(pol.action_encoder, TransformerActionEncoder) and pol.action_encoder.variational
    assert isinstance(pol.action_decoder, CNNActionDecoder) and pol.external_encoder and pol.external_decoder
    B, D = 12, 64
    batch, t = _batch(cfg, B, seed=5)
    noise = torch.randn(B, D, generator=torch.Generator().manual_seed(4)).to(DEV)
    pol.train()
    loss, metrics = pol.compute_loss(batch, t=t, enc_noise=noise)
    loss.backward()
    assert {"loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss", "action_kl_loss"} <= set(metrics)
    sd64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in pol.state_dict().items()
            if v.dtype == torch.float32 and not k.startswith("action_encoder.")}
    enc64 = TransformerActionEncoder(enc_hidden_dim=64, latent_dim=D, num_heads=cfg.policy.action_ae.net.num_heads, pred_horizon=16,
                                     action_dim=cfg.task.action_dim, num_layers=2, variational=True).double()
    enc64.load_state_dict({k: v.double().cpu() for k, v in pol.action_encoder.state_dict().items()})
    obs = pol.observer(batch).double().cpu()
    actions = batch[cfg.task.action_key].double().cpu()
    mean, logvar = torch.chunk(enc64(actions), 2, dim=1)
    z1 = mean + torch.exp(0.5 * logvar) * noise.double().cpu()
    kl = (0.5 * torch.sum(mean ** 2 + torch.exp(logvar) - 1.0 - logvar, dim=1)).mean()
    HID = pol.action_decoder.hidden_dim
    part, ref_m = O.compute_loss(sd64, obs, actions, t.double().cpu(), 6, enc_latents=z1, decoder=lambda z: O.cnn_action_decode(sd64, z, 16, HID))
    ref_loss = part + 1e-2 * kl
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    assert abs(metrics["action_kl_loss"] - kl.item()) <= 2e-5 * max(1.0, abs(kl.item()))
    for n, p in pol.action_encoder.named_parameters():
        ref = dict(enc64.named_parameters())[n].grad
        err = ((p.grad.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
        assert err <= 1e-4, (n, err)
    for k, p in pol.named_parameters():
        if p.grad is None or k.startswith("action_encoder."):
            continue
        ref = sd64[k].grad
        err = ((p.grad.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
        assert err <= 1e-4, (k, err)
    # eval mode: deterministic posterior, KL = 0 (dists.py:12-18,25-26)
    pol.eval()
    with torch.no_grad():
        le, me = pol.compute_loss(batch, t=t)
    assert me["action_kl_loss"] == 0.0
    part_e, _ = O.compute_loss({k: v.detach() for k, v in sd64.items()}, obs, actions, t.double().cpu(), 6, enc_latents=mean.detach(),
                               decoder=lambda z: O.cnn_action_decode({k: v.detach() for k, v in sd64.items()}, z, 16, HID))
    assert abs(le.item() - part_e.item()) <= 1e-5 * abs(part_e.item())
