"""An action encoder that runs as a torch module beside the fused step (VITA_ENC_EXTERNAL: its output goes into
vita_compute_loss_f32 as the latents, d loss / d latents comes back and autograd continues into the module).
1. The hook itself: a torch re-statement of SimpleActionEncoder (action_ae.py:111-170) plugged in as `action_encoder=` must
   give the losses and every gradient of the policy whose encoder runs inside the fused call.
2. `encoder_type: transformer` (vita_amd.encoders.TransformerActionEncoder): one training step against plain torch autograd of
   the same composite loss built from the oracle's pieces."""
import pytest
import torch
import torch.nn as nn

from oracle import vita_oracle as O
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cfg(encoder_type):
    cfg = default_config("robomimic_square")
    cfg.policy.vita.latent_dim = 64
    cfg.policy.action_ae.net.merge({"latent_dim": 64, "enc_hidden_dim": 64, "dec_hidden_dim": 48, "encoder_type": encoder_type, "num_layers": 2})
    cfg.policy.flow_net.merge({"hidden_dim": 64, "num_layers": 2})
    cfg.policy.flow_matcher.name = "conditional"
    return cfg


def _stats(cfg):
    s, a = cfg.task.state_dim, cfg.task.action_dim
    return {cfg.task.state_key: {"mean": [0.0] * s, "std": [1.0] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [1.0] * a}}


def _batch(cfg, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {cfg.task.state_key: torch.randn(B, 1, cfg.task.state_dim, generator=g).to(DEV),
            "observation.features": torch.relu(torch.randn(B, 1, 512, generator=g)).to(DEV),
            cfg.task.action_key: torch.randn(B, 16, cfg.task.action_dim, generator=g).to(DEV)}, torch.rand(B, generator=g).to(DEV)


class TorchSimpleEncoder(nn.Module):
    """SimpleActionEncoder in plain torch: per-step Linear(A -> D / T), num_layers x Mlp(exact GELU), flatten."""

    def __init__(self, A, T, D, NL):
        super().__init__()
        c = D // T
        self.input_proj = nn.Linear(A, c)
        self.fc1 = nn.ModuleList([nn.Linear(c, 4 * c) for _ in range(NL)])
        self.fc2 = nn.ModuleList([nn.Linear(4 * c, c) for _ in range(NL)])

    def forward(self, actions):
        x = self.input_proj(actions)
        for f1, f2 in zip(self.fc1, self.fc2):
            x = f2(torch.nn.functional.gelu(f1(x)))
        return x.reshape(actions.shape[0], -1)


def test_external_encoder_equals_the_fused_one():
    cfg = _cfg("simple")
    torch.manual_seed(0)
    native = VitaPolicy(cfg, _stats(cfg)).to(DEV)
    A, T, D, NL = cfg.task.action_dim, 16, 64, 2
    ext_mod = TorchSimpleEncoder(A, T, D, NL).to(DEV)
    nsd = native.state_dict()
    with torch.no_grad():
        ext_mod.input_proj.weight.copy_(nsd["action_encoder.input_proj.weight"]); ext_mod.input_proj.bias.copy_(nsd["action_encoder.input_proj.bias"])
        for i in range(NL):
            ext_mod.fc1[i].weight.copy_(nsd[f"action_encoder.layers.{i}.fc1.weight"]); ext_mod.fc1[i].bias.copy_(nsd[f"action_encoder.layers.{i}.fc1.bias"])
            ext_mod.fc2[i].weight.copy_(nsd[f"action_encoder.layers.{i}.fc2.weight"]); ext_mod.fc2[i].bias.copy_(nsd[f"action_encoder.layers.{i}.fc2.bias"])
    torch.manual_seed(0)
    ext = VitaPolicy(cfg, _stats(cfg), action_encoder=ext_mod).to(DEV)
    ext.load_state_dict({k: v for k, v in nsd.items() if not k.startswith("action_encoder.")}, strict=False)
    assert ext.external_encoder and not any(k.startswith("action_encoder.") for k, _ in ext.hot_path_named_parameters())
    batch, t = _batch(cfg, 24)
    native.train(); ext.train()
    ln, mn = native.compute_loss(batch, t=t)
    le, me = ext.compute_loss(batch, t=t)
    ln.backward(); le.backward()
    assert set(mn) == set(me)
    for k in mn:
        assert abs(mn[k] - me[k]) <= 2e-5 * max(1.0, abs(mn[k])), (k, mn[k], me[k])
    gn = {k: p.grad for k, p in native.named_parameters() if p.grad is not None}
    ge = {k: p.grad for k, p in ext.named_parameters() if p.grad is not None}
    for k, g in gn.items():
        if k.startswith("action_encoder."):
            continue
        err = ((g - ge[k]).abs().max() / g.abs().max().clamp_min(1e-20)).item()
        assert err <= 2e-5, (k, err)
    pairs = [("action_encoder.input_proj.weight", ext_mod.input_proj.weight), ("action_encoder.input_proj.bias", ext_mod.input_proj.bias)]
    for i in range(NL):
        pairs += [(f"action_encoder.layers.{i}.fc1.weight", ext_mod.fc1[i].weight), (f"action_encoder.layers.{i}.fc1.bias", ext_mod.fc1[i].bias),
                  (f"action_encoder.layers.{i}.fc2.weight", ext_mod.fc2[i].weight), (f"action_encoder.layers.{i}.fc2.bias", ext_mod.fc2[i].bias)]
    for k, p in pairs:      # the module's own gradients, through d loss / d latents
        err = ((gn[k] - p.grad).abs().max() / gn[k].abs().max().clamp_min(1e-20)).item()
        assert err <= 5e-5, (k, err)
    # frozen encoder: no gradient reaches the module, the losses stay the fused path's
    ext.zero_grad(set_to_none=True)
    cfg2 = _cfg("simple"); cfg2.policy.action_ae.freeze_encoder = True
    extf = VitaPolicy(cfg2, _stats(cfg2), action_encoder=ext_mod).to(DEV)
    extf.load_state_dict({k: v for k, v in nsd.items() if not k.startswith("action_encoder.")}, strict=False)
    extf.train()
    lf, _ = extf.compute_loss(batch, t=t)
    lf.backward()
    assert all(p.grad is None for p in ext_mod.parameters())
    # the native trainer declines
    with pytest.raises(NotImplementedError):
        ext.loss_and_grads(batch, t=t)


def test_transformer_encoder_policy_trains_and_matches_autograd():
    cfg = _cfg("transformer")
    torch.manual_seed(1)
    pol = VitaPolicy(cfg, _stats(cfg)).to(DEV)
    from vita_amd.encoders import TransformerActionEncoder
    assert isinstance(pol.action_encoder, TransformerActionEncoder) and pol.external_encoder
    batch, t = _batch(cfg, 16, seed=2)
    pol.train()
    loss, metrics = pol.compute_loss(batch, t=t)
    loss.backward()
    assert {"loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss"} <= set(metrics)
    # the same composite loss in fp64 torch autograd: the oracle's pieces over the encoder's fp64 copy
    sd64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in pol.state_dict().items() if v.dtype == torch.float32 and not k.startswith("action_encoder.")}
    enc64 = TransformerActionEncoder(enc_hidden_dim=64, latent_dim=64, num_heads=cfg.policy.action_ae.net.num_heads, pred_horizon=16,
                                     action_dim=cfg.task.action_dim, num_layers=2).double()
    enc64.load_state_dict({k: v.double().cpu() for k, v in pol.action_encoder.state_dict().items()})
    obs = pol.observer(batch).double().cpu()
    actions = batch[cfg.task.action_key].double().cpu()
    ref_loss, _ = O.compute_loss(sd64, obs, actions, t.double().cpu(), 6, enc_latents=enc64(actions))
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    for n, p in pol.action_encoder.named_parameters():
        ref = dict(enc64.named_parameters())[n].grad
        err = ((p.grad.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
        assert err <= 1e-4, (n, err)
    for k in ("flow_net.layers.0.mlp.fc1.weight", "action_decoder.output_proj.weight", "obs_encoder.weight"):
        ref = sd64[k].grad
        g = dict(pol.named_parameters())[k].grad.double().cpu()
        assert ((g - ref).abs().max() / ref.abs().max()).item() <= 2e-5, k
    # one optimizer step through the reference's factories moves the encoder too
    opt = pol.get_optimizer()
    before = pol.action_encoder.latent_output_proj.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, pol.action_encoder.latent_output_proj.weight)
