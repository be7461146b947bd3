"""The oracle (oracle/vita_oracle.py) against the committed golden vectors, which were
produced by the reference's own modules (tools/gen_golden.py).  CPU only.

Equality is exact when the accumulation order is the same as at generation time (same torch
build, one thread); across thread counts MKL may re-block a GEMM, so the tolerance is a few
fp32 ulps of the output scale rather than bitwise."""
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vita_oracle as O


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
def misc(golden_dir):
    return load(os.path.join(golden_dir, "misc.safetensors"))


@pytest.fixture(scope="module")
def alt(golden_dir):
    return load(os.path.join(golden_dir, "alt.safetensors"))


@pytest.fixture(scope="module")
def full(golden_dir):
    return load(os.path.join(golden_dir, "full.safetensors"))


def close(a, b, tol=2e-6):
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"normalised max error {err:g} > {tol:g}"


CASES = ["tiny_A2_cnn", "tiny_A7_cnn", "tiny_A21_cnn", "tiny_A7_simple"]


def case_sd(tiny, tag):
    p = tag + "/sd/"
    return {k[len(p):]: v for k, v in tiny.items() if k.startswith(p)}


@pytest.mark.parametrize("tag", CASES)
def test_flow_net_and_euler(tiny, tag):
    sd = case_sd(tiny, tag)
    x0, t = tiny[tag + "/x0"], tiny[tag + "/t"]
    close(O.flow_net_time_embed(sd, t), tiny[tag + "/time_embed"])
    close(O.flow_net_forward(sd, x0, t), tiny[tag + "/flow_v"])
    te = O.flow_net_time_embed(sd, t)
    for l in range(O.flow_net_num_layers(sd)):
        g, sc, sh = O.flow_net_modulation(sd, te, l)
        close(torch.cat([g, sc, sh], 1), tiny[f"{tag}/mod{l}"])
    for K in (1, 2, 6, 17):
        close(O.euler_sample(sd, x0, K), tiny[f"{tag}/euler_K{K}"])
    _, (traj, _) = O.euler_sample(sd, x0, 6, return_traces=True)
    close(torch.stack(traj), tiny[tag + "/euler_K6_traj"])


@pytest.mark.parametrize("tag", CASES)
def test_autoencoder_and_generate(tiny, tag):
    sd = case_sd(tiny, tag)
    acts = tiny[tag + "/actions"]
    T, A = acts.shape[1], acts.shape[2]
    z = O.action_encode(sd, acts)
    close(z, tiny[tag + "/enc"])
    close(O.simple_action_decode(sd, tiny[tag + "/enc"], T, A), tiny[tag + "/dec"])
    close(O.obs_encode(sd, tiny[tag + "/obs_features"]), tiny[tag + "/obs_latents"])
    close(O.generate_actions(sd, tiny[tag + "/obs_features"], 6, T, A), tiny[tag + "/generate_actions_K6"])


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("pairing", ["id", "perm"])
def test_compute_loss_and_grads(tiny, tag, pairing):
    sd = {k: v.clone().requires_grad_(True) for k, v in case_sd(tiny, tag).items()}
    of = tiny[tag + "/obs_features"].clone().requires_grad_(True)
    pj = None if pairing == "id" else tiny[tag + "/perm"]
    loss, met = O.compute_loss(sd, of, tiny[tag + "/actions"], tiny[tag + "/t"], 6, pair_j=pj)
    for k in ("flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss", "loss"):
        close(met[k].detach().reshape(1), tiny[f"{tag}/loss_{pairing}/{k}"], 1e-5)
    loss.backward()
    close(of.grad, tiny[f"{tag}/grad_{pairing}/obs_features"], 1e-4)
    if pairing == "id":
        for k, v in sd.items():
            close(v.grad, tiny[f"{tag}/grad_id/{k}"], 1e-4)


def test_misc(misc):
    close(O.sinusoidal_pos_embed(misc["misc/sin_t"], 256), misc["misc/sin_e"])
    n = "misc/normalize/"
    close(O.normalize(misc[n + "state"], misc[n + "state_mean"], misc[n + "state_std"]), misc[n + "norm_state"])
    close(O.normalize(misc[n + "act"], misc[n + "act_mean"], misc[n + "act_std"]), misc[n + "norm_act"])
    close(O.unnormalize(misc[n + "norm_act"], misc[n + "act_mean"], misc[n + "act_std"]), misc[n + "unnorm_act"])
    close(O.contrastive_loss(misc["misc/nce/a"], misc["misc/nce/b"]).reshape(1), misc["misc/nce/loss"])
    mean, logvar = torch.chunk(misc["misc/kl/params"], 2, dim=1)
    close(O.gaussian_kl(mean, logvar).reshape(1), misc["misc/kl/loss"])


@pytest.mark.parametrize("name,Ks", [("pusht_B64", (1, 6)), ("hook_package_B64", (1, 6)),
                                     ("robomimic_square_B32", (1, 6, 25))])
def test_full_dims(full, name, Ks):
    """Default dims (D=H=512, L=4): weights regenerated from seed 0, inputs from seed 1;
    the fixture holds head/tail rows and fp64 row sums of the reference's outputs."""
    obs_dim, A, B = (int(v) for v in full[name + "/meta"])
    sd = O.make_state_dict(obs_dim, A, seed=0)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(B, obs_dim, generator=g)
    for K in Ks:
        acts = O.generate_actions(sd, obs, K, 16, A).reshape(B, -1)
        close(acts[:4], full[f"{name}/K{K}/head"], 5e-6)
        close(acts[-4:], full[f"{name}/K{K}/tail"], 5e-6)
        close(acts.double().sum(1), full[f"{name}/K{K}/rowsum64"], 5e-6)


def test_fp32_oracle_close_to_fp64_truth():
    """Bounds the fp32 error of the reference algorithm itself: the 1e-5 parity budget of the
    north star has to sit above this."""
    sd = O.make_state_dict(533, 21, seed=0)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(16, 533, generator=g)
    a32 = O.generate_actions(sd, obs, 6, 16, 21)
    a64 = O.generate_actions(O.cast_sd(sd, torch.float64), obs.double(), 6, 16, 21)
    err = (a32.double() - a64).abs().max().item() / a64.abs().max().item()
    assert err < 1e-5, err


def test_alternate_matchers_inference(alt):
    """SimpleMeanFlowNet.forward, MeanFlowMatcher.sample and ConsistencyFlowMatcher.sample (both branches) against the
    reference's own outputs (tools/gen_golden_alt.py)."""
    sd = {k[len("mf/sd/"):]: v for k, v in alt.items() if k.startswith("mf/sd/")}
    u, v, _ = O.mean_flow_net_forward(sd, alt["mf/x"], alt["mf/t"], alt["mf/h"])
    close(u, alt["mf/u"])
    close(v, alt["mf/v"])
    for K in (1, 2, 5):
        close(O.mean_flow_sample(sd, alt["mf/x"], K), alt[f"mf/sample/K{K}"])
    sd = {k[len("cfm/sd/"):]: v for k, v in alt.items() if k.startswith("cfm/sd/")}
    for K in (1, 4):
        close(O.consistency_sample(sd, alt["cfm/x0"], K, sigma_var=0.0), alt[f"cfm/det/K{K}"])
        close(O.consistency_sample(sd, alt["cfm/x0"], K, noise=alt[f"cfm/sde/noise{K}"]), alt[f"cfm/sde/K{K}"])


def test_vae_branch_losses_and_gradients_match_reference_autograd(golden_dir):
    """Variational encoder branch (action_vae.py:179-229, dists.py, vita_policy.py:120-121,146-150): the oracle against
    the losses and gradients the reference's own modules + autograd produced (tools/gen_golden_vae.py)."""
    v = load(os.path.join(golden_dir, "vae.safetensors"))
    tag = "vae_A7"
    p = tag + "/sd/"
    sd = {k[len(p):]: x for k, x in v.items() if k.startswith(p)}
    obs, actions, t, eps = (v[f"{tag}/{k}"] for k in ("obs_features", "actions", "t", "eps"))
    mean, logvar, z = O.vae_cnn_action_encode(sd, actions, eps)
    close(mean, v[tag + "/mean"]); close(logvar, v[tag + "/logvar"]); close(z, v[tag + "/z"])
    for name, klw, training in (("kl1e-4", 1e-4, True), ("kl0.1", 0.1, True), ("eval", 0.1, False)):
        sdg = {k: x.clone().requires_grad_(True) for k, x in sd.items()}
        loss, m = O.compute_loss(sdg, obs, actions, t, 6, kl_weight=klw, enc_noise=eps if training else None)
        for k in m:
            close(m[k].detach().reshape(1), v[f"{tag}/{name}/loss/{k}"], 3e-6)
        keys = list(sdg)
        gs = torch.autograd.grad(loss, [sdg[k] for k in keys])
        for k, g in zip(keys, gs):
            close(g, v[f"{tag}/{name}/grad/{k}"], 2e-5)


def test_meanflow_training_losses_and_gradients_match_reference_autograd(golden_dir):
    """MeanFlow training (mean_flow_matcher.py:109-199 inside vita_policy.py:112-189): the oracle against the losses and
    gradients of the reference's own MeanFlowMatcher / SimpleMeanFlowNet / encoder / decoder + autograd
    (tools/gen_golden_meanflow.py), K = 1 and 3 sampler steps."""
    v = load(os.path.join(golden_dir, "meanflow_train.safetensors"))
    sd = {k[len("mf/sd/"):]: x for k, x in v.items() if k.startswith("mf/sd/")}
    obs, actions, t, r = (v["mf/" + k] for k in ("obs_features", "actions", "t", "r"))
    # the last two cases carry the dispersive term (dispersive_loss_weight, dispersive_loss_tau; mean_flow_matcher.py:27-39, 185-195)
    for K, use_imf, dw, tau in ((1, False, 0.0, 1.0), (3, False, 0.0, 1.0), (1, True, 0.0, 1.0), (2, True, 0.0, 1.0),
                                (1, False, 0.5, 1.0), (2, True, 0.25, 0.7)):
        tag = f"mf/{'imf_' if use_imf else ''}{'disp_' if dw > 0 else ''}K{K}"
        sdg = {k: x.clone().requires_grad_(True) for k, x in sd.items()}
        loss, m = O.compute_loss_meanflow(sdg, obs, actions, t, r, K, use_imf=use_imf, dispersive_loss_weight=dw, dispersive_loss_tau=tau)
        assert ("dispersive_loss" in m) == (dw > 0)
        for k in m:
            close(m[k].detach().reshape(1), v[f"{tag}/loss/{k}"], 3e-6)
        keys = list(sdg)
        gs = torch.autograd.grad(loss, [sdg[k] for k in keys], allow_unused=True)
        for k, g in zip(keys, gs):
            ref = v[f"{tag}/grad/{k}"]
            if g is None:
                assert ref.abs().max().item() == 0.0, k        # the v head takes no part without use_imf
            else:
                close(g, ref, 2e-5)
