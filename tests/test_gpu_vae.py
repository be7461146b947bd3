"""GPU parity of the variational action-encoder branch (policy.action_ae.use_variational; action_vae.py:179-229,
models/dists.py, vita_policy.py:120-121,146-150) through vita_compute_loss_f32, against tests/golden/vae.safetensors:
losses and every gradient from the reference's own modules + torch autograd (tools/gen_golden_vae.py).
Tolerances: losses 1e-5 relative, gradients 2e-5 normalised max error."""
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vita_oracle as O

pytestmark = pytest.mark.gpu
TAG = "vae_A7"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from vita_amd import _lib, ops as _ops
    _lib.load()
    return _ops


@pytest.fixture(scope="module")
def vae(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "vae.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def sd_of(vae):
    p = TAG + "/sd/"
    return {k[len(p):]: v for k, v in vae.items() if k.startswith(p)}


def run(ops, dev, sd, obs, actions, t, cfg, eps):
    sdd = {k: v.to(dev).contiguous() for k, v in sd.items()}
    gsd = {k: torch.full_like(v, float("nan")) for k, v in sdd.items()}
    pw = ops.PolicyHandle(sdd, dev, pred_horizon=actions.shape[1], strict=True)
    gw = ops.PolicyHandle(gsd, dev, pred_horizon=actions.shape[1], strict=True)
    assert pw.w.enc.variational == 1
    losses, dobs = ops.compute_loss(pw, cfg, obs.to(dev), actions.to(dev), t.to(dev), grads=gw, need_dobs=True,
                                    enc_noise=eps.to(dev) if eps is not None else None)
    torch.cuda.synchronize()
    return losses.cpu(), {k: v.cpu() for k, v in gsd.items()}, dobs.cpu()


KEYS = {"loss": 0, "flow_loss": 1, "consistency_loss": 2, "flow_action_recon_loss": 3, "enc_action_recon_loss": 4, "action_kl_loss": 7}


@pytest.mark.parametrize("name,klw,training", [("kl1e-4", 1e-4, True), ("kl0.1", 0.1, True), ("eval", 0.1, False)])
def test_vae_compute_loss_golden_reference_autograd(ops, dev, vae, name, klw, training):
    sd = sd_of(vae)
    obs, actions, t, eps = (vae[f"{TAG}/{k}"] for k in ("obs_features", "actions", "t", "eps"))
    losses, grads, dobs = run(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=6, kl_weight=klw), eps if training else None)
    for k, i in KEYS.items():
        ref = vae[f"{TAG}/{name}/loss/{k}"].item()
        assert abs(losses[i].item() - ref) <= 1e-5 * abs(ref) + 1e-9, f"{name} {k}: {losses[i].item()} vs {ref}"
    for k in sd:
        ref = vae[f"{TAG}/{name}/grad/{k}"]
        assert torch.isfinite(grads[k]).all(), k
        assert nerr(grads[k], ref) <= 2e-5, f"{name} grad {k}: {nerr(grads[k], ref):g}"
    assert nerr(dobs, vae[f"{TAG}/{name}/grad/obs_features"]) <= 2e-5


def test_vae_posterior_parameters_and_sample(ops, dev, vae):
    sd = sd_of(vae)
    pw = ops.PolicyHandle(sd, dev)
    params = ops.action_encoder_forward(pw, vae[TAG + "/actions"].to(dev))
    D = vae[TAG + "/mean"].shape[1]
    assert tuple(params.shape) == (5, 2 * D)
    assert nerr(params[:, :D], vae[TAG + "/mean"]) <= 1e-5 and nerr(params[:, D:], vae[TAG + "/logvar"]) <= 1e-5


def test_vae_branches_vs_oracle_fp64_and_rejections(ops, dev):
    """freeze_encoder (no KL term, :146), kl_weight 0, l2 + repeated pairings; enc_noise on a plain encoder is an error."""
    from vita_amd._lib import VitaHipError
    A, T, B, D = 7, 16, 13, 64
    sd = O.make_state_dict(40, A, latent_dim=D, hidden_dim=64, num_layers=2, pred_horizon=T, ae_hidden=48, ae_layers=4, time_embed_dim=32,
                           seed=9, variational=True)
    g = torch.Generator().manual_seed(10)
    obs, actions, t = torch.randn(B, 40, generator=g), torch.randn(B, T, A, generator=g), torch.rand(B, generator=g)
    eps = torch.randn(B, D, generator=g)
    for kw, ck in ((dict(freeze_encoder=True, kl_weight=0.3), dict(freeze_encoder=True, kl_weight=0.3)),
                   (dict(kl_weight=0.0), dict(kl_weight=0.0)),
                   (dict(kl_weight=0.5, recon_loss_type="l2"), dict(kl_weight=0.5, recon="l2"))):
        sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        loss, m = O.compute_loss(sd64, obs.double(), actions.double(), t.double(), 6, enc_noise=eps.double(), **kw)
        keys = list(sd64)
        gs = torch.autograd.grad(loss, [sd64[k] for k in keys], allow_unused=True)
        losses, grads, _ = run(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=6, **ck), eps)
        assert abs(losses[0].item() - loss.item()) <= 1e-5 * abs(loss.item()), kw
        if "action_kl_loss" in m:
            assert abs(losses[7].item() - m["action_kl_loss"].item()) <= 1e-5 * abs(m["action_kl_loss"].item())
        else:
            assert losses[7].item() == 0.0
        for k, gr in zip(keys, gs):
            if gr is None or gr.abs().max().item() == 0:
                assert grads[k].abs().max().item() == 0, f"{kw}: grad {k} should be zero"
            else:
                assert nerr(grads[k], gr) <= 2e-5, f"{kw}: grad {k} {nerr(grads[k], gr):g}"
    plain = O.make_state_dict(40, A, latent_dim=D, hidden_dim=64, num_layers=2, pred_horizon=T, ae_hidden=48, ae_layers=4, time_embed_dim=32, seed=9)
    pw = ops.PolicyHandle(plain, dev)
    with pytest.raises(VitaHipError):
        ops.compute_loss(pw, ops.loss_config(6), obs.to(dev), actions.to(dev), t.to(dev), enc_noise=eps.to(dev))


def test_policy_use_variational(dev):
    """policy.action_ae.use_variational through VitaPolicy.compute_loss: state-dict shape, metric keys, training vs eval."""
    from test_gpu_policy import small_cfg, make_policy, make_batch, hot_sd, oracle_features
    cfg = small_cfg("cnn")
    cfg.policy.action_ae.use_variational = True
    cfg.policy.action_ae.kl_weight = 1e-2
    pol = make_policy(cfg, seed=3).to(dev)
    assert tuple(pol.action_encoder.latent_proj.weight.shape)[0] == 2 * 64
    g = torch.Generator().manual_seed(4)
    B = 9
    batch = make_batch(cfg, B, g)
    t = torch.rand(B, generator=g)
    eps = torch.randn(B, 64, generator=g)
    feats, acts = oracle_features(pol, batch, cfg)
    sd64 = {k: v.double() for k, v in hot_sd(pol).items()}
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    pol.train()
    loss, met = pol.compute_loss(pol.normalize_targets(pol.normalize_inputs(dict(dbatch))), t=t.to(dev), enc_noise=eps.to(dev))
    ref, m = O.compute_loss(sd64, feats.double(), acts.double(), t.double(), 6, kl_weight=1e-2, enc_noise=eps.double())
    assert set(met) == set(m) and abs(met["action_kl_loss"] - m["action_kl_loss"].item()) <= 1e-5 * abs(m["action_kl_loss"].item())
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    loss.backward()
    assert pol.action_encoder.latent_proj.weight.grad is not None
    pol.eval()
    with torch.no_grad():
        loss_e, met_e = pol.compute_loss(pol.normalize_targets(pol.normalize_inputs(dict(dbatch))), t=t.to(dev))
    ref_e, m_e = O.compute_loss(sd64, feats.double(), acts.double(), t.double(), 6, kl_weight=1e-2, enc_noise=None)
    assert met_e["action_kl_loss"] == 0.0 and abs(loss_e.item() - ref_e.item()) <= 1e-5 * abs(ref_e.item())
