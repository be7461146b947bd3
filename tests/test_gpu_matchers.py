"""GPU parity of the torchcfm-backed matchers beyond the default (flow_matchers.py:82-99): conditional / exact with
sigma > 0, target, schrodinger.  torchcfm==1.0.5 is absent from the image and the reference holds no fixture for it, so
these paths are "parity UNPINNED": the checker is the oracle's restatement of the published formulas
(oracle.cfm_sample_location_and_conditional_flow), differentiated in fp64.  Tolerances: xt / ut 2e-6, losses 1e-5,
gradients 2e-5 (normalised max error)."""
import pytest
import torch

from oracle import vita_oracle as O

pytestmark = pytest.mark.gpu
KINDS = {"conditional": 0, "target": 1, "schrodinger": 2}


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


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("kind,sigma", [("conditional", 0.3), ("target", 0.0), ("target", 0.2), ("schrodinger", 0.5), ("schrodinger", 1.0)])
def test_fm_interpolate_forward_and_backward(ops, dev, kind, sigma):
    g = torch.Generator().manual_seed(5)
    B, D = 33, 64
    x0 = torch.randn(B, D, generator=g).double().requires_grad_(True)
    x1 = torch.randn(B, D, generator=g).double().requires_grad_(True)
    t = torch.rand(B, generator=g) * 0.9 + 0.05         # away from the bridge's / target's singular ends
    eps = torch.randn(B, D, generator=g)
    dxt, dut = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    pj, pi = torch.randint(0, B, (B,), generator=g), torch.randint(0, B, (B,), generator=g)
    for a, b in ((None, None), (pi, pj)):
        _, xt, ut = O.cfm_sample_location_and_conditional_flow(x0 if a is None else x0[a], x1 if b is None else x1[b], t.double(), sigma,
                                                               eps.double(), kind)
        g0, g1 = torch.autograd.grad((xt * dxt.double()).sum() + (ut * dut.double()).sum(), (x0, x1), allow_unused=True)
        xtd, utd = ops.fm_interpolate(KINDS[kind], sigma, x0.detach().float().to(dev), x1.detach().float().to(dev), t.to(dev), eps.to(dev),
                                      pair_j=b, pair_i=a)
        assert nerr(xtd, xt) <= 2e-6 and nerr(utd, ut) <= 5e-6, (kind, sigma, nerr(xtd, xt), nerr(utd, ut))
        d0, d1 = ops.fm_interpolate_bwd(KINDS[kind], sigma, dxt.to(dev), dut.to(dev), t.to(dev), pair_j=b, pair_i=a)
        if g0 is None:
            assert d0.abs().max().item() == 0.0        # target: x0 only lends its shape
        else:
            assert nerr(d0, g0) <= 5e-6
        assert nerr(d1, g1) <= 5e-6


def test_fm_interpolate_rejects_bad_arguments(ops, dev):
    from vita_amd._lib import VitaHipError
    x = torch.zeros(4, 8, device=dev)
    t = torch.full((4,), 0.5, device=dev)
    with pytest.raises(VitaHipError):
        ops.fm_interpolate(2, 0.0, x, x, t, x)          # Sigma must be strictly positive
    with pytest.raises(VitaHipError):
        ops.fm_interpolate(1, 0.0, x, x, t, None)       # target needs eps
    with pytest.raises(VitaHipError):
        ops.fm_interpolate(7, 0.0, x, x, t, x)
    xt, ut = ops.fm_interpolate(0, 0.0, x + 1, x + 3, t, None)
    assert torch.equal(xt, torch.full_like(x, 2.0)) and torch.equal(ut, torch.full_like(x, 2.0))


@pytest.mark.parametrize("kind,sigma", [("conditional", 0.25), ("target", 0.0), ("schrodinger", 0.6)])
def test_compute_loss_with_matcher_vs_oracle_fp64(ops, dev, kind, sigma):
    A, T, B, D = 7, 16, 17, 64
    sd = O.make_state_dict(40, A, latent_dim=D, hidden_dim=64, num_layers=2, pred_horizon=T, ae_hidden=48, ae_layers=4, time_embed_dim=32, seed=12)
    g = torch.Generator().manual_seed(13)
    obs, actions = torch.randn(B, 40, generator=g), torch.randn(B, T, A, generator=g)
    t = torch.rand(B, generator=g) * 0.9 + 0.05
    eps = torch.randn(B, D, generator=g)
    pj = torch.randperm(B, generator=g) if kind == "schrodinger" else None
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    loss, m = O.compute_loss(sd64, obs.double(), actions.double(), t.double(), 6, pair_j=pj, sigma=sigma, fm_kind=kind, fm_noise=eps.double())
    keys = list(sd64)
    gs = torch.autograd.grad(loss, [sd64[k] for k in keys])
    sdd = {k: v.to(dev).contiguous() for k, v in sd.items()}
    gsd = {k: torch.full_like(v, float("nan")) for k, v in sdd.items()}
    pw = ops.PolicyHandle(sdd, dev, pred_horizon=T, strict=True)
    gw = ops.PolicyHandle(gsd, dev, pred_horizon=T, strict=True)
    cfg = ops.loss_config(num_steps=6, fm_kind=KINDS[kind], fm_sigma=sigma)
    losses, _ = ops.compute_loss(pw, cfg, obs.to(dev), actions.to(dev), t.to(dev), grads=gw, pair_j=pj, fm_noise=eps.to(dev))
    assert abs(losses[0].item() - loss.item()) <= 1e-5 * abs(loss.item())
    assert abs(losses[1].item() - m["flow_loss"].item()) <= 1e-5 * abs(m["flow_loss"].item())
    for k, gr in zip(keys, gs):
        assert nerr(gsd[k], gr) <= 2e-5, f"{kind}: grad {k} {nerr(gsd[k], gr):g}"
    with pytest.raises(Exception):
        ops.compute_loss(pw, cfg, obs.to(dev), actions.to(dev), t.to(dev))      # the matcher's noise is required


def test_policy_with_target_and_schrodinger_matchers(dev):
    from test_gpu_policy import small_cfg, make_policy, make_batch
    for name, sigma in (("target", 0.0), ("schrodinger", 0.5), ("exact", 0.1)):
        cfg = small_cfg("cnn", matcher=name)
        cfg.policy.flow_matcher.sigma = sigma
        pol = make_policy(cfg, seed=5).to(dev)
        g = torch.Generator().manual_seed(6)
        batch = {k: v.to(dev) for k, v in make_batch(cfg, 12, g).items()}
        torch.manual_seed(0)
        loss, met = pol.forward(dict(batch))
        assert torch.isfinite(loss) and set(met) >= {"loss", "flow_loss", "consistency_loss"}
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for _, p in pol.hot_path_named_parameters())
