"""Few-row GEMM kernel (vita_gemm_f32_small.hip: M <= 512, one 32x32 tile per workgroup, in-workgroup split of the reduction,
grid-level split above K = 1024) through vita_linear_f32, against fp64 matmul: forward with every epilogue option, the
input-gradient form (column B operand, act'(z) in place), ragged M / N, K tails, determinism.  Tolerance 3e-6 normalised."""
import pytest
import torch
import torch.nn.functional as F

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


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


ACTS = {0: lambda v: v, 1: torch.relu, 2: F.gelu, 3: lambda v: F.gelu(v, approximate="tanh"), 4: F.silu, 5: F.mish}


@pytest.mark.parametrize("M,N,K", [(128, 512, 512), (256, 2048, 512), (128, 512, 2048), (25, 512, 512), (1, 32, 64), (130, 336, 512),
                                   (261, 1024, 256), (48, 2048, 512), (512, 512, 512), (96, 100, 1028), (37, 512, 4096), (128, 1536, 512)])
def test_small_forward_all_epilogues(ops, dev, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gate = torch.randn(N, generator=g)
    gmat = torch.randn(M, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    z = x.double() @ w.double().T + b.double()
    for act in (0, 2, 3, 5):
        y = ops.linear(xd, wd, b.to(dev), act)
        assert nerr(y, ACTS[act](z)) <= 3e-6, (act, nerr(y, ACTS[act](z)))
    y = ops.linear(xd, wd, None, 0)
    assert nerr(y, x.double() @ w.double().T) <= 3e-6
    y = ops.linear(xd, wd, b.to(dev), 4, gate=gate.to(dev), resid=resid.to(dev))
    assert nerr(y, resid.double() + gate.double() * F.silu(z)) <= 3e-6
    y = ops.linear(xd, wd, b.to(dev), 0, gate=gmat.to(dev))
    assert nerr(y, gmat.double() * z) <= 3e-6
    y = ops.linear(xd, wd, b.to(dev), 0, gate_scalar=1.0 / 6, resid=resid.to(dev))
    assert nerr(y, resid.double() + z / 6) <= 3e-6
    y1, y2 = ops.linear(xd, wd, b.to(dev), 3), ops.linear(xd, wd, b.to(dev), 3)
    assert torch.equal(y1, y2), "not deterministic"
    # a row's result does not depend on the other rows of the batch
    if M >= 4:
        ys = ops.linear(xd[1:3].contiguous(), wd, b.to(dev), 3)
        assert nerr(ys, y1[1:3]) <= 2e-6


@pytest.mark.parametrize("M,IN,OUT", [(128, 512, 2048), (128, 2048, 512), (256, 512, 512), (25, 512, 512), (70, 96, 132), (130, 533, 512),
                                      (128, 336, 512), (9, 64, 3072)])
def test_small_dgrad_with_activation_derivative(ops, dev, M, IN, OUT):
    g = torch.Generator().manual_seed(M + IN + OUT)
    w = torch.randn(OUT, IN, generator=g) / IN ** 0.5
    dy = torch.randn(M, OUT, generator=g)
    dx = ops.linear_t(dy.to(dev), w.to(dev), b_col=True, M=M, N=IN, K=OUT)
    assert nerr(dx, dy.double() @ w.double()) <= 3e-6
    resid = torch.randn(M, IN, generator=g)
    dx = ops.linear_t(dy.to(dev), w.to(dev), b_col=True, M=M, N=IN, K=OUT, resid=resid.to(dev), gate_scalar=0.25)
    assert nerr(dx, resid.double() + 0.25 * (dy.double() @ w.double())) <= 3e-6
    for act in (2, 3):
        z = (torch.randn(M, IN, generator=g) * 2).double().requires_grad_(True)
        (F.linear(ACTS[act](z), w.double()) * dy.double()).sum().backward()
        zd = z.detach().float().to(dev)
        out = ops.linear_t(dy.to(dev), w.to(dev), b_col=True, M=M, N=IN, K=OUT, act=act, act_grad_z=zd, out=zd)   # in place over z
        assert nerr(out, z.grad) <= 3e-6, (act, nerr(out, z.grad))
