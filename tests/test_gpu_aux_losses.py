"""GPU parity of the optional loss terms: symmetric InfoNCE (contrastive_loss.py:5-22) and the diagonal-Gaussian KL
(dists.py:23-31), forward against the reference's golden values and the oracle, backward against fp64 autograd of the
same formula (tolerance: normalised max error 1e-5 like every fp32 result here; 1e-4 for losses per the north star)."""
import os

import pytest
import torch
import torch.nn.functional as F
from safetensors import safe_open

from oracle import vita_oracle as O

pytestmark = pytest.mark.gpu


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
def misc(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "misc.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def nce64(a, b, tau=0.07):
    a, b = F.normalize(a, dim=1), F.normalize(b, dim=1)
    logits = a @ b.t() / tau
    lab = torch.arange(a.shape[0])
    return (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab)) / 2


def test_contrastive_golden(ops, dev, misc):
    a, b, want = misc["misc/nce/a"], misc["misc/nce/b"], misc["misc/nce/loss"]
    got = ops.contrastive_loss(a.to(dev), b.to(dev))
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    assert abs(got.item() - float(O.contrastive_loss(a, b))) <= 1e-5 * abs(want.item())


@pytest.mark.parametrize("B,D,tau", [(1, 8, 0.07), (2, 3, 0.5), (6, 32, 0.07), (128, 512, 0.07), (257, 130, 0.1), (1024, 512, 0.07)])
def test_contrastive_fwd_bwd(ops, dev, B, D, tau):
    g = torch.Generator().manual_seed(B * 131 + D)
    a = torch.randn(B, D, generator=g)
    b = (0.5 * a + torch.randn(B, D, generator=g)) * 3.0        # correlated pairs, un-normalised scale
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = nce64(a64, b64, tau)
    gout = torch.tensor(0.37)
    (ref * gout.double()).backward()
    loss, ga, gb = ops.contrastive_loss(a.to(dev), b.to(dev), tau, gout=gout.to(dev))
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    if B > 1:
        assert nerr(ga, a64.grad) <= 2e-5, nerr(ga, a64.grad)
        assert nerr(gb, b64.grad) <= 2e-5, nerr(gb, b64.grad)
    else:   # a single pair: the loss is log(1) = 0 and has no gradient
        assert ga.abs().max().item() <= 1e-6 and gb.abs().max().item() <= 1e-6
    # forward-only call agrees bitwise with the forward of the fwd+bwd call, and strided inputs are accepted
    wide = torch.zeros(B, D + 5, device=dev)
    wide[:, :D] = a.to(dev)
    again = ops.contrastive_loss(wide[:, :D], b.to(dev), tau)
    assert again.item() == loss.item()


def test_contrastive_zero_row(ops, dev):
    # F.normalize clamps the norm at 1e-12: an all-zero row stays zero and gets a finite gradient
    a = torch.randn(8, 16, generator=torch.Generator().manual_seed(1))
    b = torch.randn(8, 16, generator=torch.Generator().manual_seed(2))
    a[3] = 0.0
    want = nce64(a.double(), b.double())
    loss, ga, gb = ops.contrastive_loss(a.to(dev), b.to(dev), gout=torch.ones((), device=dev))
    assert abs(loss.item() - want.item()) <= 1e-5 * want.item()
    assert torch.isfinite(ga).all() and torch.isfinite(gb).all()


def test_contrastive_errors(ops, dev):
    from vita_amd._lib import VitaHipError
    with pytest.raises(VitaHipError):
        ops.contrastive_loss(torch.zeros(4, 8, device=dev), torch.zeros(4, 9, device=dev))
    with pytest.raises(VitaHipError):
        ops.contrastive_loss(torch.zeros(4, 8, device=dev), torch.zeros(4, 8, device=dev), temperature=0.0)
    with pytest.raises(VitaHipError):
        ops.contrastive_loss(torch.zeros(4, 8), torch.zeros(4, 8))


def test_kl_golden(ops, dev, misc):
    params, want = misc["misc/kl/params"], misc["misc/kl/loss"]
    got = ops.gaussian_kl(params.to(dev))
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())


@pytest.mark.parametrize("B,C", [(1, 1), (6, 32), (128, 512), (1000, 77)])
def test_kl_fwd_bwd(ops, dev, B, C):
    g = torch.Generator().manual_seed(B + C)
    p = torch.randn(B, 2 * C, generator=g)
    p64 = p.double().requires_grad_(True)
    mean, logvar = torch.chunk(p64, 2, dim=1)
    ref = (0.5 * torch.sum(mean.pow(2) + logvar.exp() - 1.0 - logvar, dim=1)).mean()
    gout = torch.tensor(1.7)
    (ref * gout.double()).backward()
    loss, gp = ops.gaussian_kl(p.to(dev), gout=gout.to(dev))
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert nerr(gp, p64.grad) <= 1e-5
    assert abs(loss.item() - float(O.gaussian_kl(*torch.chunk(p, 2, dim=1)))) <= 1e-5 * max(1.0, abs(ref.item()))
