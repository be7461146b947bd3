"""GPU parity: the HIP path (through the C ABI, via vita_amd.ops) against the oracle and the
committed golden vectors of the reference.  Run on the GPU box with `-m gpu`.

Tolerance (north star): final actions within 1e-5 relative at fp32.  "Relative" is the
normalised max error  max|a-b| / max|b|  over the tensor: fp32 GEMMs with a different
summation order cannot agree element-wise-relative on entries near zero.  The fp32 reference
itself sits ~1e-6 from the fp64 truth on this metric (test_oracle_golden.py).
"""
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vita_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def dev():
    return _gpu()


@pytest.fixture(scope="module")
def ops():
    from vita_amd import ops as _ops
    from vita_amd import _lib
    _lib.load()  # fails loudly if the HIP library is missing
    return _ops


def load(path):
    out = {}
    with safe_open(path, framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


@pytest.fixture(scope="module")
def tiny(golden_dir):
    return load(os.path.join(golden_dir, "tiny.safetensors"))


@pytest.fixture(scope="module")
def full(golden_dir):
    return load(os.path.join(golden_dir, "full.safetensors"))


@pytest.fixture(scope="module")
def misc(golden_dir):
    return load(os.path.join(golden_dir, "misc.safetensors"))


def nerr(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def close(a, b, tol=TOL, what=""):
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite output"
    e = nerr(a, b)
    assert e <= tol, f"{what}: normalised max error {e:g} > {tol:g}"


def case_sd(tiny, tag):
    p = tag + "/sd/"
    return {k[len(p):]: v for k, v in tiny.items() if k.startswith(p)}


# ----------------------------------------------------------------------------------------
# fused linear: shapes, tails, every activation / prologue / epilogue
# ----------------------------------------------------------------------------------------
ACTS = {0: lambda x: x, 1: torch.relu, 2: torch.nn.functional.gelu,
        3: lambda x: torch.nn.functional.gelu(x, approximate="tanh"),
        4: torch.nn.functional.silu, 5: torch.nn.functional.mish}


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 32, 32), (64, 128, 32), (65, 129, 33), (25, 336, 512),
                                   (130, 512, 533), (257, 2048, 512), (192, 512, 2048), (4096, 512, 512),
                                   (6, 1536, 512), (1024, 2048, 512)])
@pytest.mark.parametrize("act", [0, 3])
def test_linear_shapes(ops, dev, M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = ACTS[act](torch.nn.functional.linear(x.double(), w.double(), b.double()))
    y = ops.linear(x.to(dev), w.to(dev), b.to(dev), act)
    close(y, ref, 2e-6, f"linear {M}x{N}x{K} act{act}")


@pytest.mark.parametrize("act", [1, 2, 4, 5])
def test_linear_activations(ops, dev, act):
    g = torch.Generator().manual_seed(act)
    x = torch.randn(70, 96, generator=g) * 3
    w = torch.randn(130, 96, generator=g) / 5
    ref = ACTS[act](torch.nn.functional.linear(x.double(), w.double()))
    y = ops.linear(x.to(dev), w.to(dev), None, act)
    close(y, ref, 2e-6, f"act{act}")


def test_linear_strided_rows_and_no_bias(ops, dev):
    g = torch.Generator().manual_seed(3)
    big = torch.randn(50, 200, generator=g)
    x = big[:, 8:8 + 64]            # row stride 200, 16B-aligned start
    x2 = big[:, 3:3 + 64]           # misaligned start -> scalar load path
    w = torch.randn(40, 64, generator=g)
    for xs in (x, x2):
        ref = torch.nn.functional.linear(xs.double(), w.double())
        bd = big.to(dev)
        xd = bd[:, 8:72] if xs is x else bd[:, 3:67]
        close(ops.linear(xd, w.to(dev)), ref, 2e-6, "strided")


@pytest.mark.parametrize("per_row", [False, True])
@pytest.mark.parametrize("M,K,N", [(5, 32, 128), (100, 512, 2048), (64, 100, 40)])
def test_linear_ln_mod_prologue_and_gated_residual(ops, dev, per_row, M, K, N):
    """fc1 with LN+modulation prologue (simple_flow_net.py:44-45) then fc2 with gate+residual (:46)."""
    from vita_amd import _lib as L
    g = torch.Generator().manual_seed(11 + M)
    x = torch.randn(M, K, generator=g) * 2 + 0.5
    w1 = torch.randn(N, K, generator=g) / K ** 0.5
    b1 = torch.randn(N, generator=g) * 0.1
    w2 = torch.randn(K, N, generator=g) / N ** 0.5
    b2 = torch.randn(K, generator=g) * 0.1
    shp = (M, K) if per_row else (K,)
    scale, shift, gamma = (torch.randn(*shp, generator=g) * 0.3 for _ in range(3))
    xd = x.double()
    xn = torch.nn.functional.layer_norm(xd, (K,), None, None, 1e-6) * (scale.double() + 1) + shift.double()
    u_ref = torch.nn.functional.gelu(torch.nn.functional.linear(xn, w1.double(), b1.double()), approximate="tanh")
    y_ref = xd + torch.nn.functional.linear(u_ref, w2.double(), b2.double()) * gamma.double()
    u = ops.linear(x.to(dev), w1.to(dev), b1.to(dev), L.ACT_GELU_TANH, pro=L.PRO_LN_MOD, ln_eps=1e-6,
                   pro_mul=scale.to(dev), pro_add=shift.to(dev))
    close(u, u_ref, 3e-6, "ln_mod fc1")
    h = x.to(dev).clone()
    y = ops.linear(u, w2.to(dev), b2.to(dev), L.ACT_NONE, gate=gamma.to(dev), resid=h, out=h)  # in place on the residual
    close(y, y_ref, 3e-6, "gated residual fc2")


def test_linear_ln_affine_and_euler_epilogue(ops, dev):
    from vita_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    M, K, N = 37, 512, 512
    h = torch.randn(M, K, generator=g) * 3 - 1
    x = torch.randn(M, N, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    lw = 1 + 0.1 * torch.randn(K, generator=g)
    lb = 0.1 * torch.randn(K, generator=g)
    dt = 1.0 / 6
    v = torch.nn.functional.linear(torch.nn.functional.layer_norm(h.double(), (K,), lw.double(), lb.double(), 1e-5),
                                   w.double(), b.double())
    ref = x.double() + v * dt
    y = ops.linear(h.to(dev), w.to(dev), b.to(dev), L.ACT_NONE, pro=L.PRO_LN_AFFINE, ln_eps=1e-5, pro_mul=lw.to(dev),
                   pro_add=lb.to(dev), gate_scalar=dt, resid=x.to(dev))
    close(y, ref, 2e-6, "ln_affine + euler")


@pytest.mark.parametrize("M,N,K", [(25, 512, 2048), (128, 2048, 512), (128, 512, 2048), (1, 336, 512), (300, 512, 520), (64, 130, 4096)])
def test_linear_split_k_small_batches(ops, dev, M, N, K):
    """Serving / per-GPU training batch sizes: few output tiles and a long reduction take the split-K path (partials summed
    in a fixed order by a second kernel that applies the epilogue)."""
    from vita_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    gamma = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    z = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = r.double() + torch.nn.functional.gelu(z, approximate="tanh") * gamma.double()
    rd = r.to(dev)
    y = ops.linear(x.to(dev), w.to(dev), b.to(dev), L.ACT_GELU_TANH, gate=gamma.to(dev), resid=rd, out=rd)   # in place on the residual
    close(y, ref, 3e-6, "split-K fused epilogue")
    y2 = ops.linear(x.to(dev), w.to(dev), b.to(dev), L.ACT_GELU_TANH, gate=gamma.to(dev), resid=r.to(dev))
    assert torch.equal(y, y2), "split-K result not reproducible"


def test_linear_rejects_bad_arguments(ops, dev):
    from vita_amd._lib import VitaHipError
    x = torch.zeros(4, 8, device=dev)
    w = torch.zeros(3, 9, device=dev)
    with pytest.raises(VitaHipError):
        ops.linear(x, w)
    with pytest.raises(VitaHipError):
        ops.linear(x.cpu(), torch.zeros(3, 8))
    with pytest.raises(VitaHipError):
        ops.linear(x.double(), torch.zeros(3, 8, device=dev).double())
    y = ops.linear(torch.zeros(0, 8, device=dev), torch.zeros(3, 8, device=dev))  # empty batch is a no-op
    assert y.shape == (0, 3)


# ----------------------------------------------------------------------------------------
# reference golden vectors (tiny dims, full tensors)
# ----------------------------------------------------------------------------------------
CASES = ["tiny_A2_cnn", "tiny_A7_cnn", "tiny_A21_cnn", "tiny_A7_simple"]


@pytest.mark.parametrize("tag", CASES)
def test_golden_flow_net_and_euler(ops, dev, tiny, tag):
    sd = case_sd(tiny, tag)
    fw = ops.FlowNetHandle(sd, dev)
    x0, t = tiny[tag + "/x0"].to(dev), tiny[tag + "/t"].to(dev)
    close(ops.flownet_forward(fw, x0, t), tiny[tag + "/flow_v"], TOL, "flow_v")
    for K in (1, 2, 6, 17):
        close(ops.ode_sample(fw, x0, K), tiny[f"{tag}/euler_K{K}"], TOL, f"euler K={K}")
    out, traj = ops.ode_sample(fw, x0, 6, return_traces=True)
    close(traj, tiny[tag + "/euler_K6_traj"], TOL, "traj")
    close(out, tiny[tag + "/euler_K6"], TOL, "traj final")


@pytest.mark.parametrize("tag", CASES)
def test_golden_decoder_and_generate_actions(ops, dev, tiny, tag):
    sd = case_sd(tiny, tag)
    fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
    acts = tiny[tag + "/actions"]
    B, T, A = acts.shape
    close(ops.decoder_forward(dw, tiny[tag + "/enc"].to(dev)).view(B, T, A), tiny[tag + "/dec"], TOL, "decoder")
    out, lat = ops.generate_actions(sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev), fw, dw,
                                    tiny[tag + "/obs_features"].to(dev), 6, return_latents=True)
    close(out.view(B, T, A), tiny[tag + "/generate_actions_K6"], TOL, "generate_actions")
    close(lat, O.euler_sample(sd, tiny[tag + "/obs_latents"], 6), TOL, "latents")


@pytest.mark.parametrize("name,Ks", [("pusht_B64", (1, 6)), ("hook_package_B64", (1, 6)),
                                     ("robomimic_square_B32", (1, 6, 25))])
def test_golden_full_dims(ops, dev, full, name, Ks):
    """Default dims; the fixture holds head/tail rows + fp64 row sums of the reference output."""
    obs_dim, A, B = (int(v) for v in full[name + "/meta"])
    sd = O.make_state_dict(obs_dim, A, seed=0)
    fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(B, obs_dim, generator=g).to(dev)
    for K in Ks:
        acts = ops.generate_actions(sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev), fw, dw, obs, K)
        ref_scale = full[f"{name}/K{K}/head"].abs().max()
        for part, sl in (("head", slice(0, 4)), ("tail", slice(B - 4, B))):
            e = ((acts[sl].cpu() - full[f"{name}/K{K}/{part}"]).abs().max() / ref_scale).item()
            assert e <= TOL, f"{name} K={K} {part}: {e:g}"
        rs = acts.double().sum(1).cpu()
        e = ((rs - full[f"{name}/K{K}/rowsum64"]).abs().max() / full[f"{name}/K{K}/rowsum64"].abs().max()).item()
        assert e <= TOL, f"{name} K={K} rowsum: {e:g}"


# ----------------------------------------------------------------------------------------
# full size (BASELINE.json: hook_package, B=4096, K=6): oracle on a row sample + properties
# ----------------------------------------------------------------------------------------
def test_full_size_hook_package_b4096(ops, dev):
    B, O_, A, K = 4096, 533, 21, 6
    sd = O.make_state_dict(O_, A, seed=0)
    fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(B, O_, generator=g)
    ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
    acts = ops.generate_actions(ow, ob, fw, dw, obs.to(dev), K)
    assert torch.isfinite(acts).all()
    # (1) oracle on a sample of rows (rows are independent: no op on the path mixes rows)
    idx = torch.cat([torch.arange(0, 8), torch.arange(2040, 2056), torch.arange(4088, 4096)])
    ref = O.generate_actions(sd, obs[idx], K, 16, A).reshape(len(idx), -1)
    close(acts[idx.to(dev)], ref, TOL, "B=4096 sample rows")
    # (2) row independence: the same rows at other positions of a batch of the same size (same kernels, other tiles and
    #     other workgroups) come out bitwise identical; sub-batches of other sizes select other tile shapes / k-splits
    #     (a different summation order) and agree to fp32 rounding
    rolled = ops.generate_actions(ow, ob, fw, dw, torch.roll(obs, 1000, 0).to(dev), K)
    assert torch.equal(torch.roll(rolled, -1000, 0), acts), "rows depend on their position in the batch"
    sub = ops.generate_actions(ow, ob, fw, dw, obs[1000:4000].to(dev), K)
    close(sub, acts[1000:4000], 2e-6, "3000-row sub-batch (other tile shapes)")
    small = ops.generate_actions(ow, ob, fw, dw, obs[1000:1100].to(dev), K)
    close(small, acts[1000:1100], 2e-6, "small sub-batch (split-K path)")
    assert torch.equal(small, ops.generate_actions(ow, ob, fw, dw, obs[1000:1100].to(dev), K)), "split-K path not deterministic"
    # (3) determinism
    again = ops.generate_actions(ow, ob, fw, dw, obs.to(dev), K)
    assert torch.equal(again, acts)


def test_euler_linearity_property(ops, dev):
    """With K steps the sampler is x_K = x_0 + (1/K) sum v_k: the returned trajectory must satisfy
    traj[k+1] - traj[k] == dt * v(traj[k], k/K) where v comes from the per-row-t entry point."""
    sd = O.make_state_dict(533, 21, seed=0)
    fw = ops.FlowNetHandle(sd, dev)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(300, 512, generator=g).to(dev)
    K = 4
    out, traj = ops.ode_sample(fw, x0, K, return_traces=True)
    assert torch.equal(out, traj[K]) and torch.equal(traj[0], x0)
    for k in range(K):
        t = torch.full((300,), k / K, device=dev)
        v = ops.flownet_forward(fw, traj[k], t)
        step = traj[k] + v * (1.0 / K)
        close(traj[k + 1], step, 2e-6, f"step {k}")


@pytest.mark.parametrize("H", [1024, 2048])
def test_wide_velocity_net(ops, dev, H):
    """BASELINE configs[4]: thread_needle dims with flow_net.hidden_dim 1024 / 2048 (the weights no longer fit L2)."""
    sd = O.make_state_dict(533, 21, hidden_dim=H, seed=3)
    fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
    g = torch.Generator().manual_seed(H)
    obs = torch.randn(256, 533, generator=g)
    acts = ops.generate_actions(sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev), fw, dw, obs.to(dev), 6)
    idx = torch.tensor([0, 1, 127, 128, 254, 255])
    ref = O.generate_actions(O.cast_sd(sd, torch.float64), obs[idx].double(), 6, 16, 21).reshape(len(idx), -1)
    close(acts[idx.to(dev)], ref, TOL, f"H={H}")


def test_many_steps_k100(ops, dev):
    sd = O.make_state_dict(555, 7, seed=0)
    fw = ops.FlowNetHandle(sd, dev)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(8, 512, generator=g)
    ref = O.euler_sample(O.cast_sd(sd, torch.float64), x0.double(), 100)
    close(ops.ode_sample(fw, x0.to(dev), 100), ref, TOL, "K=100 vs fp64")


# ----------------------------------------------------------------------------------------
# losses, interpolation, normalisation, time embedding
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [(1,), (5, 7), (128, 512), (4096, 512), (4096, 16, 21), (1000003,)])
def test_losses_fwd_bwd(ops, dev, n):
    g = torch.Generator().manual_seed(len(n))
    a = torch.randn(*n, generator=g)
    b = torch.randn(*n, generator=g)
    a.view(-1)[0] = b.view(-1)[0]  # exercise sign(0) = 0
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    for name, fwd, bwd, ref_fn in (("mse", ops.mse_loss_fwd, ops.mse_loss_bwd, torch.nn.functional.mse_loss),
                                   ("l1", ops.l1_loss_fwd, ops.l1_loss_bwd, torch.nn.functional.l1_loss)):
        ref = ref_fn(ad, bd)
        ga_ref, gb_ref = torch.autograd.grad(ref * 0.5, (ad, bd))
        out = fwd(a.to(dev), b.to(dev))
        assert abs(out.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-12, name
        ga, gb = bwd(a.to(dev), b.to(dev), torch.tensor(0.5, device=dev))
        close(ga, ga_ref, 1e-6, name + " grad a")
        close(gb, gb_ref, 1e-6, name + " grad b")
        assert torch.equal(fwd(a.to(dev), b.to(dev)), out), "loss not reproducible"


def test_cfm_interpolate(ops, dev):
    g = torch.Generator().manual_seed(2)
    x0, x1, t = torch.randn(33, 512, generator=g), torch.randn(33, 512, generator=g), torch.rand(33, generator=g)
    perm = torch.randperm(33, generator=g)
    for pj in (None, perm):
        _, xt_ref, ut_ref = O.cfm_sample_location_and_conditional_flow(x0, x1 if pj is None else x1[pj], t)
        xt, ut = ops.cfm_interpolate(x0.to(dev), x1.to(dev), t.to(dev), pj)
        assert torch.equal(xt.cpu(), xt_ref) and torch.equal(ut.cpu(), ut_ref)


def test_normalize_golden(ops, dev, misc):
    n = "misc/normalize/"
    d = lambda k: misc[n + k].to(dev)  # noqa: E731
    close(ops.normalize(d("state"), d("state_mean"), d("state_std")), misc[n + "norm_state"], 1e-6)
    close(ops.normalize(d("act"), d("act_mean"), d("act_std")), misc[n + "norm_act"], 1e-6)
    close(ops.unnormalize(d("norm_act"), d("act_mean"), d("act_std")), misc[n + "unnorm_act"], 1e-6)


@pytest.mark.parametrize("M,H", [(1, 4), (5, 32), (37, 100), (300, 512), (64, 2048), (9, 33)])
@pytest.mark.parametrize("per_row", [False, True])
def test_layernorm_modulate(ops, dev, M, H, per_row):
    from vita_amd import _lib as L
    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(M, H, generator=g) * 2 + 1
    shp = (M, H) if per_row else (H,)
    mul, add = torch.randn(*shp, generator=g) * 0.3, torch.randn(*shp, generator=g) * 0.3
    xn = torch.nn.functional.layer_norm(x.double(), (H,), None, None, 1e-6)
    close(ops.layernorm_modulate(x.to(dev), mul.to(dev), add.to(dev), 1e-6, L.PRO_LN_MOD), xn * (1 + mul.double()) + add.double(), 2e-6)
    xn5 = torch.nn.functional.layer_norm(x.double(), (H,), None, None, 1e-5)
    close(ops.layernorm_modulate(x.to(dev), mul.to(dev), add.to(dev), 1e-5, L.PRO_LN_AFFINE), xn5 * mul.double() + add.double(), 2e-6)


def test_sinusoid_golden(ops, dev, misc):
    close(ops.sinusoidal_embed(misc["misc/sin_t"].to(dev), 256), misc["misc/sin_e"], 1e-6)


@pytest.mark.parametrize("B,K,traces", [(1024, 1, False), (1024, 6, True), (2048, 3, False)])
def test_folded_layernorm_path_matches_the_layernorm_kernels(ops, dev, B, K, traces):
    """At B >= 1024 (B % 256 == 0) the sampler folds the LayerNorms into the GEMM epilogues (vita_api.hip: use_ln_fold);
    halves of 512 rows take the LayerNorm-kernel path.  Same rows, two code paths: agreement to fp32 rounding, and the
    fp64 oracle on a row sample to the usual 1e-5."""
    sd = O.make_state_dict(555, 7, seed=3)
    fw = ops.FlowNetHandle(sd, dev)
    g = torch.Generator().manual_seed(B + K)
    x0 = torch.randn(B, 512, generator=g)
    if traces:
        out, traj = ops.ode_sample(fw, x0.to(dev), K, return_traces=True)
        assert torch.equal(traj[-1], out) and torch.equal(traj[0].cpu(), x0)
    else:
        out = ops.ode_sample(fw, x0.to(dev), K)
    parts = torch.cat([ops.ode_sample(fw, x0[i:i + 512].to(dev), K) for i in range(0, B, 512)])
    close(out, parts, 2e-6, "folded vs LayerNorm-kernel path")
    idx = torch.arange(0, B, B // 16)
    sd64 = {k: v.double() for k, v in sd.items() if k.startswith("flow_net.")}
    close(out[idx.to(dev)], O.euler_sample(sd64, x0[idx].double(), K), TOL, "folded path vs fp64 oracle")
    assert torch.equal(out, ops.ode_sample(fw, x0.to(dev), K)), "folded path not deterministic"
