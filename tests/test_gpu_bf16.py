"""GPU: the bf16-operand inference path.  Stated tolerance (north star: "within a stated bf16 tolerance"):
  * a single bf16 GEMM against the SAME bf16-rounded operands multiplied in fp64: 2e-6 (accumulation is fp32: the only
    difference is summation order) — this pins the kernel;
  * the whole generate_actions path against the fp32 reference semantics (oracle): 2e-2 normalised max error on the
    actions, 1e-2 on the latents (bf16 has 8 significant bits; 6 Euler steps x 4 layers of rounded operands)."""
import pytest
import torch

from oracle import vita_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_ACTION_TOL = 2e-2
BF16_LATENT_TOL = 1e-2


@pytest.fixture(scope="module")
def ops():
    from vita_amd import _lib, ops as _ops
    _lib.load()
    return _ops


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float64)


def unpack(bits, cols):
    return bits.cpu().view(torch.bfloat16)[:, :cols].to(torch.float64)


@pytest.mark.parametrize("M,N,K", [(1, 8, 8), (5, 32, 40), (130, 336, 533), (64, 128, 64), (300, 512, 2048), (4096, 512, 512), (257, 2048, 512)])
def test_linear_bf16_matches_fp64_on_the_rounded_operands(ops, M, N, K):
    from vita_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    gamma, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    xb, wb = ops.pack_bf16(x.to(DEV)), ops.pack_bf16(w.to(DEV))
    assert xb.shape == (M, (K + 7) & ~7) and torch.equal(unpack(xb, K), bf16_round(x)), "pack is not round-to-nearest-even"
    if K % 8:
        assert int(xb[:, K:].abs().max()) == 0, "padding columns must be zero"
    z = bf16_round(x) @ bf16_round(w).T + b.double()
    ref = r.double() + torch.nn.functional.gelu(z, approximate="tanh") * gamma.double()
    y, yb = ops.linear_bf16(xb, wb, b.to(DEV), L.ACT_GELU_TANH, gate=gamma.to(DEV), resid=r.to(DEV), out_bf16=True)
    assert nerr(y, ref) <= 3e-6, nerr(y, ref)
    assert torch.equal(unpack(yb, N), bf16_round(y.cpu())), "bf16 copy differs from the rounded fp32 output"
    y2, _ = ops.linear_bf16(xb, wb, b.to(DEV), L.ACT_GELU_TANH, gate=gamma.to(DEV), resid=r.to(DEV))
    assert torch.equal(y, y2)


@pytest.mark.parametrize("R,M,N,split", [(32, 8, 8, False), (100, 136, 72, False), (1000, 336, 512, True), (4096, 2048, 512, True),
                                         (28672, 512, 2048, True), (777, 128, 128, True), (5000, 512, 512, False)])
def test_linear_bf16_tn_matches_fp64_on_the_rounded_operands(ops, R, M, N, split):
    """dW = dY^T X from row-major bf16 operands (transposed LDS reads): ragged R (not a multiple of the 32-row k-tile), ragged
    M / N (multiples of 8 only), with and without the split over R.  Asymmetric random operands; tolerance 2e-6 of the result's
    scale (fp32 accumulation: only the summation order differs from fp64)."""
    g = torch.Generator().manual_seed(R + M + N)
    a = torch.randn(R, M, generator=g).to(DEV)
    b = (torch.randn(R, N, generator=g) * 0.5 + 0.1).to(DEV)
    ab, bb = ops.pack_bf16(a), ops.pack_bf16(b)
    c = ops.linear_bf16_tn(ab, bb, M, N, split=split)
    ref = bf16_round(a.cpu()).T @ bf16_round(b.cpu())
    assert c.shape == (M, N)
    assert nerr(c, ref) < 2e-6
    # a second call on the same operands: deterministic to the bit
    assert torch.equal(c, ops.linear_bf16_tn(ab, bb, M, N, split=split))


def test_linear_bf16_tn_reads_operands_inside_wider_rows(ops):
    """Operands that are column blocks of wider matrices (lda > M): the stacked activation buffers of the training step."""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(640, 256, generator=g).to(DEV)
    b = torch.randn(640, 384, generator=g).to(DEV)
    ab, bb = ops.pack_bf16(a), ops.pack_bf16(b)
    c = ops.linear_bf16_tn(ab[:, 64:192], bb[:, 128:256], 128, 128)
    ref = bf16_round(a.cpu())[:, 64:192].T @ bf16_round(b.cpu())[:, 128:256]
    assert nerr(c, ref) < 2e-6


def test_linear_bf16_tn_rejects_ragged_widths(ops):
    from vita_amd import _lib as L
    ab = torch.zeros(64, 16, dtype=torch.int16, device=DEV)
    with pytest.raises(L.VitaHipError):
        ops.linear_bf16_tn(ab, ab, 12, 16)


def test_linear_bf16_rejects_unpacked_operands(ops):
    from vita_amd._lib import VitaHipError
    import ctypes as C
    from vita_amd import _lib as L
    lib = L.load()
    a = L.LinearArgs()
    x = torch.zeros(4, 12, dtype=torch.int16, device=DEV)
    w = torch.zeros(3, 12, dtype=torch.int16, device=DEV)
    y = torch.zeros(4, 3, device=DEV)
    a.X, a.ldx, a.W, a.ldw, a.Y, a.ldy, a.M, a.N, a.K = x.data_ptr(), 12, w.data_ptr(), 12, y.data_ptr(), 3, 4, 3, 12
    assert lib.vita_linear_bf16(C.byref(a), None, 0, None) == -1 and b"multiples of 8" in lib.vita_last_error()
    with pytest.raises(VitaHipError):
        ops.linear_bf16(torch.zeros(4, 16, dtype=torch.int16, device=DEV), torch.zeros(3, 8, dtype=torch.int16, device=DEV))


@pytest.mark.parametrize("task,obs_dim,A,B", [("hook_package", 533, 21, 300), ("pusht", 514, 2, 64), ("robomimic_square", 555, 7, 25)])
def test_generate_actions_bf16_within_stated_tolerance(ops, task, obs_dim, A, B):
    sd = O.make_state_dict(obs_dim, A, seed=0)
    fw, dw, bw = ops.FlowNetHandle(sd, DEV), ops.DecoderHandle(sd, DEV), ops.Bf16Weights(sd, DEV)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(B, obs_dim, generator=g)
    acts, lat = ops.generate_actions_bf16(sd["obs_encoder.bias"].to(DEV), fw, dw, bw, obs.to(DEV), 6, return_latents=True)
    ref_lat = O.euler_sample(sd, O.obs_encode(sd, obs), 6)
    ref = O.simple_action_decode(sd, ref_lat, 16, A).reshape(B, -1)
    e_lat, e_act = nerr(lat, ref_lat), nerr(acts, ref)
    assert torch.isfinite(acts).all()
    assert e_lat <= BF16_LATENT_TOL, f"{task}: latent error {e_lat:g}"
    assert e_act <= BF16_ACTION_TOL, f"{task}: action error {e_act:g}"
    # and it really is a different arithmetic from the fp32 path (guards against silently running fp32)
    f32 = ops.generate_actions(sd["obs_encoder.weight"].to(DEV), sd["obs_encoder.bias"].to(DEV), fw, dw, obs.to(DEV), 6)
    assert nerr(acts, f32) > 1e-5
    again = ops.generate_actions_bf16(sd["obs_encoder.bias"].to(DEV), fw, dw, bw, obs.to(DEV), 6)
    assert torch.equal(again, acts)


def test_generate_actions_bf16_b4096_rows_independent(ops):
    sd = O.make_state_dict(533, 21, seed=0)
    fw, dw, bw = ops.FlowNetHandle(sd, DEV), ops.DecoderHandle(sd, DEV), ops.Bf16Weights(sd, DEV)
    g = torch.Generator().manual_seed(2)
    obs = torch.randn(4096, 533, generator=g).to(DEV)
    ob = sd["obs_encoder.bias"].to(DEV)
    acts = ops.generate_actions_bf16(ob, fw, dw, bw, obs, 6)
    # a row's result does not depend on its position in a batch: the same rows as the second half of a batch of 8192
    # (both sizes take the persistent launch: same tiles, same bits)
    other = torch.randn(4096, 533, generator=g).to(DEV)
    big = ops.generate_actions_bf16(ob, fw, dw, bw, torch.cat([other, obs]), 6)
    assert torch.equal(big[4096:], acts)
    # other batch sizes take other kernels (2048 rows: per-launch GEMMs with the folded LayerNorm; 3000 rows: the LayerNorm
    # kernels): equal up to bf16 noise
    for lo, hi in ((1024, 3072), (1000, 4000)):
        sub = ops.generate_actions_bf16(ob, fw, dw, bw, obs[lo:hi], 6)
        assert nerr(sub, acts[lo:hi]) <= BF16_ACTION_TOL, (lo, hi)
    idx = torch.arange(0, 4096, 512)
    ref = O.generate_actions(sd, obs[idx].cpu(), 6, 16, 21).reshape(len(idx), -1)
    assert nerr(acts[idx.to(DEV)], ref) <= BF16_ACTION_TOL


def test_policy_gemm_dtype_bf16_repacks_after_a_weight_update(ops):
    from vita_amd.config import default_config
    from vita_amd.policy import VitaPolicy
    cfg = default_config("pusht")
    cfg.policy.gemm_dtype = "bf16"
    stats = {"observation.state": {"mean": [0.0, 0.0], "std": [1.0, 1.0]}, "action": {"mean": [0.0, 0.0], "std": [1.0, 1.0]}}
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats).to(DEV)
    g = torch.Generator().manual_seed(3)
    batch = {"observation.state": torch.randn(9, 1, 2, generator=g).to(DEV), "observation.features": torch.randn(9, 1, 512, generator=g).to(DEV)}
    a1 = pol.generate_actions(batch)
    sd = {k: p.detach().cpu() for k, p in pol.hot_path_named_parameters()}
    feats = torch.cat([batch["observation.state"].flatten(1), batch["observation.features"].flatten(1)], 1).cpu()
    ref = O.generate_actions(sd, feats, 6, 16, 2)
    assert 1e-6 < nerr(a1, ref) <= BF16_ACTION_TOL
    with torch.no_grad():
        pol.action_decoder.output_proj.weight.mul_(2.0)      # an "optimizer step": the cached bf16 copy is now stale
    a2 = pol.generate_actions(batch)
    sd["action_decoder.output_proj.weight"] = sd["action_decoder.output_proj.weight"] * 2.0
    assert nerr(a2, O.generate_actions(sd, feats, 6, 16, 2)) <= BF16_ACTION_TOL
    cfg.policy.gemm_dtype = "fp8"
    with pytest.raises(ValueError):
        VitaPolicy(cfg, stats)
