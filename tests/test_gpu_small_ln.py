"""LayerNorm (+ modulation) as a prologue of the few-row GEMM (gemm_f32_small_lds_kernel<false, 8, true>: M <= 512, K = 512, one
modulation row for the batch; per-row tables take the generic kernel's prologue): against fp64 torch on the same inputs — vector and per-row modulation, LN_MOD and LN_AFFINE, ragged M and N, every epilogue the fc1 /
out_proj calls use (bias + tanh-GELU + saved pre-activation; bias + dt * v + residual) — and
against the separate LayerNorm kernel + plain GEMM it replaces (same arithmetic, another summation order: 2e-6)."""
import pytest
import torch
import torch.nn.functional as F

from vita_amd import _lib as L
from vita_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _ref(x, w, b, mul, add, eps, mod):
    x, w, b, mul, add = (t.double().cpu() for t in (x, w, b, mul, add))
    n = F.layer_norm(x, (x.shape[1],), eps=eps)
    a = n * ((1.0 + mul) if mod else mul) + add
    return a, a @ w.T + b


@pytest.mark.parametrize("M,N", [(128, 2048), (256, 2048), (25, 512), (77, 336), (512, 512), (1, 64)])
@pytest.mark.parametrize("per_row", [False, True])
def test_layernorm_prologue_of_the_few_row_gemm(M, N, per_row):
    g = torch.Generator().manual_seed(M * 7 + N)
    K = 512
    x = (torch.randn(M, K, generator=g) * 1.7 + 0.3).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    shape = (M, K) if per_row else (K,)
    mul = (torch.randn(shape, generator=g) * 0.2).to(DEV)
    add = (torch.randn(shape, generator=g) * 0.2).to(DEV)
    # fc1 form: LN_MOD, bias + tanh-GELU
    y = ops.linear(x, w, b, L.ACT_GELU_TANH, pro=L.PRO_LN_MOD, ln_eps=1e-6, pro_mul=mul, pro_add=add)
    a_ref, z_ref = _ref(x, w, b, mul, add, 1e-6, True)
    y_ref = F.gelu(z_ref, approximate="tanh")
    assert (y.double().cpu() - y_ref).abs().max().item() <= 5e-6 * max(1.0, y_ref.abs().max().item())
    # the kernels it replaces: LayerNorm launch + plain GEMM
    hn = ops.layernorm_modulate(x, mul, add, eps=1e-6, mode=L.PRO_LN_MOD)
    y2 = ops.linear(hn, w, b, L.ACT_GELU_TANH)
    assert (y - y2).abs().max().item() <= 2e-6 * max(1.0, y2.abs().max().item())
    # out_proj form: LN_AFFINE, bias, Euler update on a residual
    if not per_row:
        r = torch.randn(M, N, generator=g).to(DEV)
        y3 = ops.linear(x, w, b, L.ACT_NONE, pro=L.PRO_LN_AFFINE, ln_eps=1e-5, pro_mul=mul, pro_add=add, gate_scalar=1.0 / 6.0, resid=r)
        _, z3 = _ref(x, w, b, mul, add, 1e-5, False)
        ref3 = r.double().cpu() + z3 / 6.0
        assert (y3.double().cpu() - ref3).abs().max().item() <= 5e-6 * max(1.0, ref3.abs().max().item())
