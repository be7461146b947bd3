#!/usr/bin/env python3
"""Generate tests/golden/*.safetensors from the REFERENCE's own modules.

Runs only in the build container (needs /root/reference); only its output fixtures travel.
The reference files are loaded by path (importlib) so that `flare/policies/__init__.py`
(which needs diffusers) is bypassed.  Two third-party imports the image lacks are satisfied
by in-memory modules, for import resolution only:
  * timm.layers.Mlp   -> restated from timm==1.0.20 (fc1 -> act -> drop1 -> norm -> fc2 -> drop2);
    this op order is therefore "restated, unpinned" (see oracle/vita_oracle.py header);
  * torchcfm.conditional_flow_matching -> empty placeholder classes; nothing captured here
    calls into them (TorchFlowMatcher.sample is pure torch; compute_loss's CFM draw is
    restated in the oracle and unpinned).
Inputs (weights, x0, t, actions) come from oracle.make_state_dict / seeded generators, so the
GPU box can regenerate them without the reference; outputs come from the reference modules.

The script also asserts, at every captured point, that the oracle restatement equals the
reference output bit-for-bit on this CPU (same torch build, same op sequence).
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
from safetensors.torch import save_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import vita_oracle as O  # noqa: E402


# -- import shims (resolution only) -------------------------------------------------------
class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


timm = types.ModuleType("timm")
timm_layers = types.ModuleType("timm.layers")
timm_layers.Mlp = _Mlp
timm.layers = timm_layers
sys.modules["timm"] = timm
sys.modules["timm.layers"] = timm_layers

tcfm = types.ModuleType("torchcfm")
tcfm_c = types.ModuleType("torchcfm.conditional_flow_matching")
for _n in ("ConditionalFlowMatcher", "TargetConditionalFlowMatcher",
           "SchrodingerBridgeConditionalFlowMatcher", "ExactOptimalTransportConditionalFlowMatcher"):
    setattr(tcfm_c, _n, type(_n, (), {"__init__": lambda self, **kw: None}))
tcfm.conditional_flow_matching = tcfm_c
sys.modules["torchcfm"] = tcfm
sys.modules["torchcfm.conditional_flow_matching"] = tcfm_c


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


sfn = load_by_path("ref_simple_flow_net", "flare/policies/vita/simple_flow_net.py")
aae = load_by_path("ref_action_ae", "flare/networks/vita/action_ae.py")
fmm = load_by_path("ref_flow_matchers", "flare/flow/flow_matchers.py")
from flare.models.pos_embed import SinusoidalPosEmbed  # noqa: E402
from flare.utils.normalize import Normalize, Unnormalize  # noqa: E402
from flare.losses.contrastive_loss import compute_contrastive_loss  # noqa: E402
from flare.models.dists import DiagonalGaussianDistribution  # noqa: E402


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def build_ref(sd, obs_dim, action_dim, D, H, L, R, T, aeh, ael, enc_type, ted=256):
    flow = sfn.SimpleFlowNet(D, H, D, L, mlp_ratio=R, dropout=0.0, time_embed_dim=ted)
    flow.load_state_dict(sub(sd, "flow_net."), strict=True)
    if enc_type == "cnn":
        enc = aae.CNNActionEncoder(pred_horizon=T, action_dim=action_dim, latent_dim=D,
                                   hidden_dim=aeh, num_layers=ael)
    else:
        enc = aae.SimpleActionEncoder(latent_dim=D, pred_horizon=T, action_dim=action_dim,
                                      num_layers=ael)
    enc.load_state_dict(sub(sd, "action_encoder."), strict=True)
    dec = aae.SimpleActionDecoder(dec_hidden_dim=aeh, latent_dim=D, pred_horizon=T,
                                  action_dim=action_dim, num_layers=ael)
    dec.load_state_dict(sub(sd, "action_decoder."), strict=True)
    obs = nn.Linear(obs_dim, D)
    obs.load_state_dict(sub(sd, "obs_encoder."), strict=True)
    for m in (flow, enc, dec, obs):
        m.double() if sd["obs_encoder.weight"].dtype == torch.float64 else m.float()
    return flow, enc, dec, obs


def ref_compute_loss(flow, enc, dec, obs, FM, obs_features, gt, t, K, pair_j=None):
    """vita_policy.py:112-189 replayed on the reference modules, with the torchcfm draw
    replaced by its restated formula and explicit (t, pairing)."""
    F = torch.nn.functional
    obs_lat = obs(obs_features)
    act_lat = enc(gt)
    x0, x1 = obs_lat, act_lat if pair_j is None else act_lat[pair_j]
    tt = t[:, None]
    xt = tt * x1 + (1 - tt) * x0
    ut = x1 - x0
    vt = flow(xt, t)
    flow_loss = torch.mean((vt - ut) ** 2)
    pred = FM.sample(flow, (obs_lat.shape[0], obs_lat.shape[1]), "cpu", start=obs_lat, num_steps=K)
    cons = F.mse_loss(pred, act_lat)
    r1 = F.l1_loss(dec(pred), gt)
    r2 = F.l1_loss(dec(act_lat), gt)
    loss = flow_loss + 1.0 * cons + 0.5 * r1 + 0.5 * r2
    return loss, dict(flow_loss=flow_loss, consistency_loss=cons, flow_action_recon_loss=r1,
                      enc_action_recon_loss=r2, loss=loss)


def eq(a, b, what):
    if not torch.equal(a, b):
        d = (a - b).abs().max().item()
        raise SystemExit(f"oracle != reference at {what}: max abs diff {d:g}")


def tiny_case(A, enc_type, out):
    D = H = 32
    L, R, T = 2, 4, 16
    aeh, ael = 32, 4 if enc_type == "cnn" else 2
    B = 5
    obs_dim = 32 + A
    tag = f"tiny_A{A}_{enc_type}"
    ted = 16  # small time-embedding width keeps the fixture small; 256 is covered by full_case
    sd = O.make_state_dict(obs_dim, A, D, H, L, R, T, aeh, ael, time_embed_dim=ted,
                           encoder_type=enc_type, seed=100 + A)
    flow, enc, dec, obs = build_ref(sd, obs_dim, A, D, H, L, R, T, aeh, ael, enc_type, ted)
    FM = fmm.TorchFlowMatcher(fm=None, num_sampling_steps=6)
    g = torch.Generator().manual_seed(7 + A)
    obs_features = torch.randn(B, obs_dim, generator=g)
    actions = torch.randn(B, T, A, generator=g)
    t = torch.rand(B, generator=g)
    x0 = torch.randn(B, D, generator=g)
    for k, v in sd.items():
        out[f"{tag}/sd/{k}"] = v
    out[f"{tag}/obs_features"] = obs_features
    out[f"{tag}/actions"] = actions
    out[f"{tag}/t"] = t
    out[f"{tag}/x0"] = x0
    with torch.no_grad():
        v = flow(x0, t)
        out[f"{tag}/flow_v"] = v
        eq(O.flow_net_forward(sd, x0, t), v, tag + " flow_v")
        te = flow.time_embed(t)
        eq(O.flow_net_time_embed(sd, t), te, tag + " time_embed")
        out[f"{tag}/time_embed"] = te
        for l in range(L):
            m = flow.layers[l].time_modulator(te)
            out[f"{tag}/mod{l}"] = m
            gam, sc, sh = O.flow_net_modulation(sd, te, l)
            eq(torch.cat([gam, sc, sh], 1), m, tag + f" mod{l}")
        for K in (1, 2, 6, 17):
            xk, (traj, vel) = FM.sample(flow, x0.shape, "cpu", num_steps=K, return_traces=True, start=x0)
            out[f"{tag}/euler_K{K}"] = xk
            eq(O.euler_sample(sd, x0, K), xk, tag + f" euler K={K}")
            if K == 6:
                out[f"{tag}/euler_K6_traj"] = torch.stack([x0] + [x for x in traj[1:]])
        z = enc(actions)
        out[f"{tag}/enc"] = z
        eq(O.action_encode(sd, actions), z, tag + " enc")
        a = dec(z)
        out[f"{tag}/dec"] = a
        eq(O.simple_action_decode(sd, z, T, A), a, tag + " dec")
        ol = obs(obs_features)
        out[f"{tag}/obs_latents"] = ol
        eq(O.obs_encode(sd, obs_features), ol, tag + " obs_enc")
        acts = dec(FM.sample(flow, ol.shape, "cpu", num_steps=6, start=ol))
        out[f"{tag}/generate_actions_K6"] = acts
        eq(O.generate_actions(sd, obs_features, 6, T, A), acts, tag + " generate_actions")
    # loss + gradients through the reference autograd (identity pairing and one permutation)
    perm = torch.tensor([2, 0, 4, 1, 3])
    out[f"{tag}/perm"] = perm
    for pname, pj in (("id", None), ("perm", perm)):
        for m in (flow, enc, dec, obs):
            m.zero_grad(set_to_none=True)
        of = obs_features.clone().requires_grad_(True)
        loss, met = ref_compute_loss(flow, enc, dec, obs, FM, of, actions, t, 6, pj)
        loss.backward()
        for k, v in met.items():
            out[f"{tag}/loss_{pname}/{k}"] = v.detach().reshape(1)
        oloss, omet = O.compute_loss(sd, obs_features, actions, t, 6, pair_j=pj)
        for k in met:
            eq(omet[k].reshape(1), met[k].detach().reshape(1), tag + f" loss_{pname}/{k}")
        out[f"{tag}/grad_{pname}/obs_features"] = of.grad
        if pname == "perm":
            continue  # weight gradients are kept for the identity pairing only (fixture size)
        for prefix, mod in (("flow_net.", flow), ("action_encoder.", enc), ("action_decoder.", dec),
                            ("obs_encoder.", obs)):
            for n, p in mod.named_parameters():
                out[f"{tag}/grad_{pname}/{prefix}{n}"] = p.grad.clone()


def full_case(name, obs_dim, A, B, out, Ks=(1, 6)):
    """Full default dims: weights regenerated from the seed on the test side; keep only
    first/last 4 rows and fp64 row-sum checksums of the reference outputs."""
    D = H = 512
    L, R, T, aeh, ael = 4, 4, 16, 512, 4
    sd = O.make_state_dict(obs_dim, A, seed=0)
    flow, enc, dec, obs = build_ref(sd, obs_dim, A, D, H, L, R, T, aeh, ael, "cnn")
    FM = fmm.TorchFlowMatcher(fm=None, num_sampling_steps=6)
    g = torch.Generator().manual_seed(1)
    obs_features = torch.randn(B, obs_dim, generator=g)
    with torch.no_grad():
        ol = obs(obs_features)
        for K in Ks:
            z = FM.sample(flow, ol.shape, "cpu", num_steps=K, start=ol)
            acts = dec(z)
            eq(O.generate_actions(sd, obs_features, K, T, A), acts, f"{name} K={K}")
            flat = acts.reshape(B, -1)
            out[f"{name}/K{K}/head"] = flat[:4].clone()
            out[f"{name}/K{K}/tail"] = flat[-4:].clone()
            out[f"{name}/K{K}/rowsum64"] = flat.double().sum(1)
            out[f"{name}/K{K}/lat_rowsum64"] = z.double().sum(1)
    out[f"{name}/meta"] = torch.tensor([obs_dim, A, B], dtype=torch.int64)


def misc_case(out):
    g = torch.Generator().manual_seed(11)
    # SinusoidalPosEmbed, pos_embed.py:11-18
    t = torch.cat([torch.rand(6, generator=g), torch.tensor([0.0, 1.0, 5 / 6])])
    e = SinusoidalPosEmbed(256)(t)
    eq(O.sinusoidal_pos_embed(t, 256), e, "sinusoid")
    out["misc/sin_t"] = t
    out["misc/sin_e"] = e
    # Normalize / Unnormalize, normalize.py:38-70
    stats = {"action": {"mean": torch.randn(7, generator=g), "std": torch.rand(7, generator=g) + 0.1},
             "observation.state": {"mean": torch.randn(9, generator=g), "std": torch.rand(9, generator=g)}}
    stats["observation.state"]["std"][3] = 0.0  # eps path
    nrm = Normalize(["observation.state"], stats)
    nrt = Normalize(["action"], stats)
    unn = Unnormalize(["action"], stats)
    state = torch.randn(4, 1, 9, generator=g)
    act = torch.randn(4, 16, 7, generator=g)
    ns = nrm({"observation.state": state})["observation.state"]
    na = nrt({"action": act})["action"]
    ua = unn({"action": na})["action"]
    eq(O.normalize(state, stats["observation.state"]["mean"], stats["observation.state"]["std"]), ns, "normalize")
    eq(O.unnormalize(na, stats["action"]["mean"], stats["action"]["std"]), ua, "unnormalize")
    for k, v in (("state", state), ("act", act), ("norm_state", ns), ("norm_act", na), ("unnorm_act", ua),
                 ("state_mean", stats["observation.state"]["mean"]), ("state_std", stats["observation.state"]["std"]),
                 ("act_mean", stats["action"]["mean"]), ("act_std", stats["action"]["std"])):
        out[f"misc/normalize/{k}"] = v
    # InfoNCE, contrastive_loss.py:5-22
    a = torch.randn(6, 32, generator=g)
    b = torch.randn(6, 32, generator=g)
    c = compute_contrastive_loss(a, b)
    eq(O.contrastive_loss(a, b), c, "contrastive")
    out["misc/nce/a"], out["misc/nce/b"], out["misc/nce/loss"] = a, b, c.reshape(1)
    # KL, dists.py:23-31
    params = torch.randn(6, 64, generator=g)
    kl = DiagonalGaussianDistribution(params).kl().mean()
    mean, logvar = torch.chunk(params, 2, dim=1)
    eq(O.gaussian_kl(mean, logvar), kl, "kl")
    out["misc/kl/params"], out["misc/kl/loss"] = params, kl.reshape(1)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)  # bit-stable accumulation order for the equality asserts
    gold = os.path.join(REPO, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    out = {}
    for A in (2, 7, 21):
        tiny_case(A, "cnn", out)
    tiny_case(7, "simple", out)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(gold, "tiny.safetensors"))
    print("tiny:", len(out), "tensors")
    out = {}
    misc_case(out)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(gold, "misc.safetensors"))
    print("misc:", len(out), "tensors")
    out = {}
    full_case("pusht_B64", 514, 2, 64, out, Ks=(1, 6))
    full_case("hook_package_B64", 533, 21, 64, out, Ks=(1, 6))
    full_case("robomimic_square_B32", 555, 7, 32, out, Ks=(1, 6, 25))
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(gold, "full.safetensors"))
    print("full:", len(out), "tensors")


if __name__ == "__main__":
    main()
