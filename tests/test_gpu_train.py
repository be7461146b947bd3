"""GPU parity of the training side: backward kernels against torch autograd (fp64 on the CPU), and the fused
vita_compute_loss_f32 (forward + backward of VitaPolicy.compute_loss) against
  * the committed golden losses and gradients produced by the reference's own modules and autograd
    (tests/golden/tiny.safetensors, tools/gen_golden.py), and
  * the oracle differentiated in fp64 at the default dims.
Tolerances: loss values 1e-4 relative (north star; measured ~1e-6), gradients 2e-5 normalised max error."""
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
def tiny(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "tiny.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def close(a, b, tol, what=""):
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite"
    e = nerr(a, b)
    assert e <= tol, f"{what}: normalised max error {e:g} > {tol:g}"


# ----------------------------------------------------------------------------------------
# transposed-operand GEMM: dgrad / wgrad / fused activation derivative
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,IN,OUT", [(1, 1, 1), (5, 32, 32), (128, 512, 2048), (130, 533, 512), (37, 105, 32), (4096, 512, 512),
                                      (896, 2048, 512), (64, 336, 512),
                                      # from 2048 stacked rows the weight gradient takes the row-major LDS-DMA kernel
                                      # (vita_gemm_f32_tn.hip): ragged R, ragged M / N (multiples of 4), the step's own shapes
                                      (2048, 2048, 512), (5000, 336, 512), (3001, 132, 68), (28672, 512, 2048), (2050, 533, 512)])
def test_dgrad_wgrad(ops, dev, M, IN, OUT):
    g = torch.Generator().manual_seed(M + IN + OUT)
    x = torch.randn(M, IN, generator=g)
    w = torch.randn(OUT, IN, generator=g) / IN ** 0.5
    dy = torch.randn(M, OUT, generator=g)
    dx = ops.linear_t(dy.to(dev), w.to(dev), b_col=True, M=M, N=IN, K=OUT)
    close(dx, dy.double() @ w.double(), 3e-6, "dgrad")
    dw = ops.linear_t(dy.to(dev), x.to(dev), a_col=True, b_col=True, M=OUT, N=IN, K=M)
    close(dw, dy.double().T @ x.double(), 3e-6, "wgrad")
    # accumulate into an existing gradient (resid = out), scaled
    acc = torch.randn(OUT, IN, generator=g)
    accd = acc.to(dev)
    ops.linear_t(dy.to(dev), x.to(dev), a_col=True, b_col=True, M=OUT, N=IN, K=M, gate_scalar=0.5, resid=accd, out=accd)
    close(accd, acc.double() + 0.5 * (dy.double().T @ x.double()), 3e-6, "wgrad accumulate")


@pytest.mark.parametrize("act", [1, 2, 3, 4, 5])
def test_dgrad_with_activation_derivative_in_place(ops, dev, act):
    fn = {1: torch.relu, 2: F.gelu, 3: lambda v: F.gelu(v, approximate="tanh"), 4: F.silu, 5: F.mish}[act]
    g = torch.Generator().manual_seed(act)
    M, IN, OUT = 70, 96, 130
    z = (torch.randn(M, IN, generator=g) * 2).double().requires_grad_(True)
    w = torch.randn(OUT, IN, generator=g) / IN ** 0.5
    dy = torch.randn(M, OUT, generator=g)
    (F.linear(fn(z), w.double()) * dy.double()).sum().backward()
    zd = z.detach().float().to(dev)
    out = ops.linear_t(dy.to(dev), w.to(dev), b_col=True, M=M, N=IN, K=OUT, act=act, act_grad_z=zd, out=zd)  # in place over z
    close(out, z.grad, 3e-6, f"dgrad*act'({act})")
    close(ops.act_bwd(z.detach().float().to(dev), torch.ones(M, IN, device=dev), act),
          torch.autograd.grad(fn(z).sum(), z)[0], 3e-6, f"act_bwd({act})")
    close(ops.act_fwd(z.detach().float().to(dev), act), fn(z.detach()), 3e-6, f"act_fwd({act})")


def test_transposed_gemm_rejects_bad_arguments(ops, dev):
    from vita_amd._lib import VitaHipError
    x = torch.zeros(8, 16, device=dev)
    with pytest.raises(VitaHipError):   # act without act_grad_z
        ops.linear_t(x, torch.zeros(16, 4, device=dev), b_col=True, M=8, N=4, K=16, act=3)
    with pytest.raises(VitaHipError):   # leading dimension too small for the claimed N
        ops.linear_t(x, torch.zeros(16, 4, device=dev), b_col=True, M=8, N=9, K=16)


# ----------------------------------------------------------------------------------------
# elementwise backward kernels
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,H", [(1, 4), (5, 32), (37, 100), (300, 512), (16, 2048)])
@pytest.mark.parametrize("per_row", [False, True])
def test_layernorm_modulate_bwd(ops, dev, M, H, per_row):
    from vita_amd import _lib as L
    g = torch.Generator().manual_seed(M * 3 + H)
    x = (torch.randn(M, H, generator=g) * 2 + 1).double().requires_grad_(True)
    shp = (M, H) if per_row else (H,)
    mul = (torch.randn(*shp, generator=g) * 0.3).double().requires_grad_(True)
    add = (torch.randn(*shp, generator=g) * 0.3).double().requires_grad_(True)
    dy = torch.randn(M, H, generator=g)
    init = torch.randn(M, H, generator=g)
    for mode, eps in ((L.PRO_LN_MOD, 1e-6), (L.PRO_LN_AFFINE, 1e-5)):
        n = F.layer_norm(x, (H,), None, None, eps)
        y = n * (1 + mul) + add if mode == L.PRO_LN_MOD else n * mul + add
        gx, gm, ga = torch.autograd.grad((y * dy.double()).sum(), (x, mul, add))
        dx, dm, da = ops.layernorm_modulate_bwd(x.detach().float().to(dev), dy.to(dev), mul.detach().float().to(dev), eps, mode)
        close(dx, gx, 5e-6, "dx")
        if per_row:
            close(dm, gm, 5e-6, "dmul rows")
            close(da, ga, 5e-6, "dadd rows")
        else:
            close(ops.colsum(dm), gm, 5e-6, "dmul")
            close(ops.colsum(da), ga, 5e-6, "dadd")
        dx2, _, _ = ops.layernorm_modulate_bwd(x.detach().float().to(dev), dy.to(dev), mul.detach().float().to(dev), eps, mode,
                                               dx_init=init.to(dev))
        close(dx2, gx + init.double(), 5e-6, "dx accumulate")


def test_gate_bwd_and_colsum(ops, dev):
    g = torch.Generator().manual_seed(3)
    M, N = 77, 96
    dy, z = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    for gate in (torch.randn(N, generator=g), torch.randn(M, N, generator=g)):
        dz, dg = ops.gate_bwd(dy.to(dev), z.to(dev), gate.to(dev))
        close(dz, dy.double() * gate.double(), 1e-6, "dz")
        close(dg, dy.double() * z.double(), 1e-6, "dgate")
    dz, _ = ops.gate_bwd(dy.to(dev), z.to(dev), 1.0 / 6, need_dgate=False)
    close(dz, dy.double() / 6, 1e-6, "scalar gate")
    for (m, n) in [(1, 1), (5, 7), (4096, 6144), (1000, 336), (63, 2048)]:
        x = torch.randn(m, n, generator=g)
        out = ops.colsum(x.to(dev))
        close(out, x.double().sum(0), 2e-6, f"colsum {m}x{n}")
        out2 = ops.colsum(x.to(dev), out=out, accumulate=True)
        close(out2, 2 * x.double().sum(0), 2e-6, "colsum accumulate")
        assert torch.equal(ops.colsum(x.to(dev)), ops.colsum(x.to(dev))), "colsum not deterministic"


@pytest.mark.parametrize("B,Lin,Cc", [(3, 16, 21), (2, 8, 32), (5, 2, 7), (1, 4, 1)])
def test_conv_k5s2_as_im2col_gemm(ops, dev, B, Lin, Cc):
    """Conv1d(k5,s2,p2) == GEMM over im2col rows with the checkpoint weight viewed (Cout, Cin*5); col2im is the adjoint."""
    g = torch.Generator().manual_seed(B + Lin + Cc)
    Cout = 24
    x = torch.randn(B, Lin, Cc, generator=g)                     # channels-last
    w = torch.randn(Cout, Cc, 5, generator=g) * 0.2
    b = torch.randn(Cout, generator=g) * 0.1
    xr = x.double().transpose(1, 2).requires_grad_(True)         # (B, C, L) as the reference feeds it
    ref = F.relu(F.conv1d(xr, w.double(), b.double(), stride=2, padding=2))   # (B, Cout, Lout)
    col = ops.im2col_k5s2(x.to(dev))
    y = ops.linear(col, w.reshape(Cout, Cc * 5).to(dev), b.to(dev), 1)        # (B*Lout, Cout) channels-last
    Lout = ref.shape[2]
    close(y.view(B, Lout, Cout), ref.detach().transpose(1, 2), 3e-6, "conv fwd")
    close(ops.nlc_ncl(y.view(B, Lout, Cout), True), ref.detach(), 3e-6, "flatten order")
    dcol = torch.randn(B * Lout, Cc * 5, generator=g)
    dx = ops.col2im_k5s2(dcol.to(dev), B, Lin, Cc)
    # adjoint identity: <im2col(x), dcol> == <x, col2im(dcol)>
    lhs = (col.cpu().double() * dcol.double()).sum()
    rhs = (x.double() * dx.cpu().double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)), (lhs, rhs)
    back = ops.nlc_ncl(ops.nlc_ncl(x.to(dev), True), False)
    assert torch.equal(back.cpu(), x)


def test_cfm_interpolate_bwd(ops, dev):
    g = torch.Generator().manual_seed(8)
    B, D = 33, 64
    x0 = torch.randn(B, D, generator=g).double().requires_grad_(True)
    x1 = torch.randn(B, D, generator=g).double().requires_grad_(True)
    t = torch.rand(B, generator=g)
    dxt, dut = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    pj = torch.randint(0, B, (B,), generator=g)   # with repeats, as the OT sampler draws them
    pi = torch.randint(0, B, (B,), generator=g)
    for a, b in ((None, None), (pi, pj)):
        _, xt, ut = O.cfm_sample_location_and_conditional_flow(x0 if a is None else x0[a], x1 if b is None else x1[b], t.double())
        g0, g1 = torch.autograd.grad((xt * dxt.double()).sum() + (ut * dut.double()).sum(), (x0, x1))
        d0, d1 = ops.cfm_interpolate_bwd(dxt.to(dev), dut.to(dev), t.to(dev), pair_j=b, pair_i=a)
        close(d0, g0, 2e-6, "dx0")
        close(d1, g1, 2e-6, "dx1")


def test_sumsq_and_fused_adamw_ema(ops, dev):
    g = torch.Generator().manual_seed(12)
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    m, v, ema = torch.zeros(n), torch.zeros(n), p.clone()
    pd, gd, md, vd, ed = (x.to(dev).clone() for x in (p, gr, m, v, ema))
    ss = ops.sumsq(gd)
    assert abs(ss.item() - (gr.double() ** 2).sum().item()) <= 1e-6 * (gr.double() ** 2).sum().item()
    lr, wd, betas, eps = 1e-3, 1e-2, (0.95, 0.999), 1e-8
    pr, mr, vr, er = p.double().clone(), m.double().clone(), v.double().clone(), ema.double().clone()
    for step in (1, 2, 3):
        norm = gr.double().norm()
        coef = min(1.0, 1.0 / (norm.item() + 1e-6))           # clip_grad_norm_(max_norm=1.0)
        O.adamw_step(pr, gr.double() * coef, mr, vr, step, lr, betas[0], betas[1], eps, wd)
        decay = O.ema_decay(step + 1)
        er.sub_((1 - decay) * (er - pr))
        ops.adamw_ema_step(pd, gd, md, vd, ed, step=step, lr=lr, betas=betas, eps=eps, weight_decay=wd, gnorm_sq=ops.sumsq(gd),
                           max_norm=1.0, ema_decay=decay)
    close(pd, pr, 2e-6, "param")
    close(md, mr, 2e-6, "exp_avg")
    close(vd, vr, 2e-6, "exp_avg_sq")
    close(ed, er, 2e-6, "ema")


# ----------------------------------------------------------------------------------------
# fused compute_loss forward + backward
# ----------------------------------------------------------------------------------------
CASES = ["tiny_A2_cnn", "tiny_A7_cnn", "tiny_A21_cnn", "tiny_A7_simple"]
LOSS_KEYS = ("loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss")
LOSS_KEYS_ALL = LOSS_KEYS + ("enc_contrastive_loss", "flow_contrastive_loss")


def case_sd(tiny, tag):
    p = tag + "/sd/"
    return {k[len(p):]: v for k, v in tiny.items() if k.startswith(p)}


def run_fused(ops, dev, sd, obs, actions, t, cfg, pair_j=None, pair_i=None, enc_noise=None):
    sdd = {k: v.to(dev).contiguous() for k, v in sd.items()}
    gsd = {k: torch.full_like(v, float("nan")) for k, v in sdd.items()}   # must be fully overwritten
    pw = ops.PolicyHandle(sdd, dev, pred_horizon=actions.shape[1], strict=True)
    gw = ops.PolicyHandle(gsd, dev, pred_horizon=actions.shape[1], strict=True)
    losses, dobs = ops.compute_loss(pw, cfg, obs.to(dev), actions.to(dev), t.to(dev), grads=gw, pair_j=pair_j, pair_i=pair_i,
                                    need_dobs=True, enc_noise=enc_noise.to(dev) if enc_noise is not None else None)
    torch.cuda.synchronize()
    return losses.cpu(), {k: v.cpu() for k, v in gsd.items()}, dobs.cpu()


@pytest.mark.parametrize("tag", CASES)
def test_compute_loss_golden_reference_autograd(ops, dev, tiny, tag):
    """Losses and every gradient against the reference's own modules + torch autograd (identity pairing), and the
    losses + d obs_features for the hand-made permutation pairing."""
    sd = case_sd(tiny, tag)
    obs, actions, t = tiny[tag + "/obs_features"], tiny[tag + "/actions"], tiny[tag + "/t"]
    cfg = ops.loss_config(num_steps=6)
    losses, grads, dobs = run_fused(ops, dev, sd, obs, actions, t, cfg)
    for i, k in enumerate(LOSS_KEYS):
        ref = tiny[f"{tag}/loss_id/{k}"].item()
        assert abs(losses[i].item() - ref) <= 1e-4 * abs(ref), f"{k}: {losses[i].item()} vs {ref}"
        assert abs(losses[i].item() - ref) <= 5e-6 * abs(ref) + 1e-7, f"{k} (tight): {losses[i].item()} vs {ref}"
    for k in sd:
        close(grads[k], tiny[f"{tag}/grad_id/{k}"], 2e-5, "grad " + k)
    close(dobs, tiny[f"{tag}/grad_id/obs_features"], 2e-5, "d obs_features")
    perm = tiny[tag + "/perm"]
    losses, grads, dobs = run_fused(ops, dev, sd, obs, actions, t, cfg, pair_j=perm)
    for i, k in enumerate(LOSS_KEYS):
        ref = tiny[f"{tag}/loss_perm/{k}"].item()
        assert abs(losses[i].item() - ref) <= 1e-4 * abs(ref), f"perm {k}: {losses[i].item()} vs {ref}"
    close(dobs, tiny[f"{tag}/grad_perm/obs_features"], 2e-5, "perm d obs_features")


def oracle_grads(sd, obs, actions, t, **kw):
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    obs64 = obs.double().requires_grad_(True)
    loss, m = O.compute_loss(sd64, obs64, actions.double(), t.double(), **kw)
    keys = list(sd64)
    gs = torch.autograd.grad(loss, [sd64[k] for k in keys] + [obs64], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(sd64[k])) for k, g in zip(keys, gs[:-1])}
    return m, grads, gs[-1]


@pytest.mark.parametrize("variant", ["default", "l2_pairs", "no_decode", "no_cons_no_enc_recon", "k1", "contrastive", "contrastive_enc_only",
                                     "contrastive_pairs_no_recon", "freeze_encoder", "freeze_decoder", "freeze_both_contrastive"])
def test_compute_loss_config_variants_vs_oracle_fp64(ops, dev, variant):
    """Small dims (D=H=64, L=3), every branch of vita_policy.py:153-187, repeated pairing indices."""
    A, T, B = 7, 16, 19
    sd = O.make_state_dict(40, A, latent_dim=64, hidden_dim=64, num_layers=3, mlp_ratio=4, pred_horizon=T, ae_hidden=48, ae_layers=4,
                           time_embed_dim=32, seed=5)
    g = torch.Generator().manual_seed(6)
    obs, actions, t = torch.randn(B, 40, generator=g), torch.randn(B, T, A, generator=g), torch.rand(B, generator=g)
    kw, ck, pi, pj = dict(num_steps=6), dict(num_steps=6), None, None
    if variant == "l2_pairs":
        kw.update(recon_loss_type="l2"); ck.update(recon="l2")
        pi, pj = torch.randint(0, B, (B,), generator=g), torch.randint(0, B, (B,), generator=g)
        kw.update(pair_i=pi, pair_j=pj)
    elif variant == "no_decode":
        kw.update(decode_flow_latents=False); ck.update(decode_flow_latents=False)
    elif variant == "no_cons_no_enc_recon":
        kw.update(consistency_weight=0.0, enc_recon_weight=0.0); ck.update(consistency_weight=0.0, enc_recon_weight=0.0)
    elif variant == "k1":
        kw.update(num_steps=1); ck.update(num_steps=1)
    elif variant == "contrastive":          # vita_policy.py:134-139 and :167-172 together
        kw.update(enc_contrastive_weight=0.3, flow_contrastive_weight=0.2); ck.update(enc_contrastive_weight=0.3, flow_contrastive_weight=0.2)
    elif variant == "contrastive_enc_only":  # no sampler: the flow term must not be computed even with a weight
        kw.update(enc_contrastive_weight=0.7, flow_contrastive_weight=0.2, decode_flow_latents=False)
        ck.update(enc_contrastive_weight=0.7, flow_contrastive_weight=0.2, decode_flow_latents=False)
    elif variant == "contrastive_pairs_no_recon":
        pi, pj = torch.randint(0, B, (B,), generator=g), torch.randint(0, B, (B,), generator=g)
        kw.update(flow_contrastive_weight=1.5, flow_recon_weight=0.0, consistency_weight=0.0, pair_i=pi, pair_j=pj)
        ck.update(flow_contrastive_weight=1.5, flow_recon_weight=0.0, consistency_weight=0.0)
    elif variant == "freeze_encoder":        # :119 no grad into the encoder, :153 no sampler, decoder still trained on dec(z1)
        kw.update(freeze_encoder=True); ck.update(freeze_encoder=True)
    elif variant == "freeze_decoder":        # :153,174,183: flow loss only, the encoder still gets the flow gradient
        kw.update(freeze_decoder=True); ck.update(freeze_decoder=True)
    elif variant == "freeze_both_contrastive":   # :142-143 early return after the flow + encoder-contrastive terms
        kw.update(freeze_encoder=True, freeze_decoder=True, enc_contrastive_weight=0.4, flow_contrastive_weight=0.4)
        ck.update(freeze_encoder=True, freeze_decoder=True, enc_contrastive_weight=0.4, flow_contrastive_weight=0.4)
    m, gref, dobs_ref = oracle_grads(sd, obs, actions, t, **kw)
    losses, grads, dobs = run_fused(ops, dev, sd, obs, actions, t, ops.loss_config(**ck), pair_j=pj, pair_i=pi)
    for i, k in enumerate(LOSS_KEYS_ALL):
        if k in m:
            assert abs(losses[i].item() - m[k].item()) <= 1e-5 * abs(m[k].item()), f"{variant} {k}"
        else:
            assert losses[i].item() == 0.0, f"{variant}: {k} should not be computed"
    for k in sd:
        scale = max(gref[k].abs().max().item(), 1e-12)
        if gref[k].abs().max().item() == 0:
            assert grads[k].abs().max().item() == 0, f"{variant}: grad {k} should be exactly zero"
        else:
            assert ((grads[k].double() - gref[k]).abs().max().item() / scale) <= 2e-5, f"{variant}: grad {k}"
    close(dobs, dobs_ref, 2e-5, "d obs_features")


def test_compute_loss_default_dims_b128_vs_oracle_fp64(ops, dev):
    """hook_package dims, per-GPU training batch 128 (default_policy.yaml:34), K=6."""
    O_, A, T, B = 533, 21, 16, 128
    sd = O.make_state_dict(O_, A, seed=0)
    g = torch.Generator().manual_seed(21)
    obs, actions, t = torch.randn(B, O_, generator=g), torch.randn(B, T, A, generator=g), torch.rand(B, generator=g)
    m, gref, dobs_ref = oracle_grads(sd, obs, actions, t, num_steps=6)
    losses, grads, dobs = run_fused(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=6))
    for i, k in enumerate(LOSS_KEYS):
        assert abs(losses[i].item() - m[k].item()) <= 1e-5 * abs(m[k].item()), f"{k}: {losses[i].item()} vs {m[k].item()}"
    worst = max((nerr(grads[k], gref[k]), k) for k in sd)
    assert worst[0] <= 2e-5, f"worst gradient error {worst}"
    close(dobs, dobs_ref, 2e-5, "d obs_features")
    # determinism (identity pairing: no atomics anywhere)
    losses2, grads2, _ = run_fused(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=6))
    assert torch.equal(losses, losses2) and all(torch.equal(grads[k], grads2[k]) for k in sd)


def test_compute_loss_large_batch_transposed_weight_dgrads_vs_oracle_fp64(ops, dev):
    """B >= 1024: the step keeps transposed copies of the weights, so every input-gradient GEMM (dX = dY . W) runs as a
    row-operand GEMM on the LDS-DMA kernels with the act'(z) epilogue.  Small feature dims keep the fp64 oracle cheap; the
    variational CNN encoder and an explicit pairing are on so that every registered weight class is exercised."""
    A, T, B = 7, 16, 1280
    sd = O.make_state_dict(40, A, latent_dim=64, hidden_dim=64, num_layers=2, mlp_ratio=4, pred_horizon=T, ae_hidden=48, ae_layers=4,
                           time_embed_dim=32, seed=15, variational=True)
    g = torch.Generator().manual_seed(16)
    obs, actions, t = torch.randn(B, 40, generator=g), torch.randn(B, T, A, generator=g), torch.rand(B, generator=g)
    noise = torch.randn(B, 64, generator=g)
    kw = dict(num_steps=3, kl_weight=0.01, enc_noise=noise.double())
    m, gref, dobs_ref = oracle_grads(sd, obs, actions, t, **kw)
    losses, grads, dobs = run_fused(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=3, kl_weight=0.01), enc_noise=noise)
    assert abs(losses[0].item() - m["loss"].item()) <= 1e-5 * abs(m["loss"].item())
    worst = max((nerr(grads[k], gref[k]), k) for k in sd)
    assert worst[0] <= 5e-5, f"worst gradient error {worst}"
    close(dobs, dobs_ref, 5e-5, "d obs_features")


def test_compute_loss_forward_only_and_action_encoder(ops, dev, tiny):
    for tag in ("tiny_A7_cnn", "tiny_A7_simple"):
        sd = case_sd(tiny, tag)
        pw = ops.PolicyHandle(sd, dev)
        acts = tiny[tag + "/actions"]
        close(ops.action_encoder_forward(pw, acts.to(dev)), tiny[tag + "/enc"], 1e-5, tag + " encoder")
        losses, dobs = ops.compute_loss(pw, ops.loss_config(6), tiny[tag + "/obs_features"].to(dev), acts.to(dev), tiny[tag + "/t"].to(dev))
        assert dobs is None
        ref = tiny[f"{tag}/loss_id/loss"].item()
        assert abs(losses[0].item() - ref) <= 1e-5 * abs(ref)


def test_compute_loss_rejects_bad_arguments(ops, dev, tiny):
    from vita_amd._lib import VitaHipError
    sd = case_sd(tiny, "tiny_A7_cnn")
    pw = ops.PolicyHandle(sd, dev)
    obs, acts, t = (tiny["tiny_A7_cnn/" + k].to(dev) for k in ("obs_features", "actions", "t"))
    with pytest.raises(VitaHipError):
        ops.compute_loss(pw, ops.loss_config(0), obs, acts, t)
    with pytest.raises(VitaHipError):
        ops.compute_loss(pw, ops.loss_config(6), obs[:, :-1], acts, t)
    with pytest.raises(VitaHipError):
        ops.compute_loss(pw, ops.loss_config(6), obs, acts[:, :8], t)
    with pytest.raises(ValueError):
        ops.loss_config(6, recon="huber")


def test_side_stream_branches_match_the_single_stream_step_bitwise(ops, dev):
    """side_streams=True (opt-in), B <= 1024: decoder weight gradients / time-path backward / encoder backward run on
    library-owned side streams.  The same kernels on the same data must give the same bits as the single-stream order, call
    after call (a race would not)."""
    O_, A, T, B = 533, 21, 16, 128
    sd = O.make_state_dict(O_, A, seed=0)
    g = torch.Generator().manual_seed(33)
    obs, actions, t = torch.randn(B, O_, generator=g), torch.randn(B, T, A, generator=g), torch.rand(B, generator=g)
    l0, g0, d0 = run_fused(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=6))
    for rep in range(4):
        l1, g1, d1 = run_fused(ops, dev, sd, obs, actions, t, ops.loss_config(num_steps=6, side_streams=True))
        assert torch.equal(l0, l1) and torch.equal(d0, d1), rep
        bad = [k for k in sd if not torch.equal(g0[k], g1[k])]
        assert not bad, (rep, bad)
    # also on a non-default stream, with the simple encoder and the contrastive terms
    sd2 = O.make_state_dict(40, 7, latent_dim=64, hidden_dim=64, num_layers=3, pred_horizon=T, ae_hidden=48, ae_layers=2, time_embed_dim=32,
                            seed=5, encoder_type="simple")
    obs2, act2, t2 = torch.randn(19, 40, generator=g), torch.randn(19, T, 7, generator=g), torch.rand(19, generator=g)
    kw = dict(num_steps=6, enc_contrastive_weight=0.3, flow_contrastive_weight=0.2)
    l0, g0, _ = run_fused(ops, dev, sd2, obs2, act2, t2, ops.loss_config(**kw))
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        l1, g1, _ = run_fused(ops, dev, sd2, obs2, act2, t2, ops.loss_config(side_streams=True, **kw))
    assert torch.equal(l0, l1) and all(torch.equal(g0[k], g1[k]) for k in sd2)
