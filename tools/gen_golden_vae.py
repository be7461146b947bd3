#!/usr/bin/env python3
"""Generate tests/golden/vae.safetensors from the REFERENCE's variational action encoder
(flare/networks/vita/action_vae.py CNNActionEncoder + flare/models/dists.py) and a replay of
VitaPolicy.compute_loss's variational branch (vita_policy.py:120-121,146-150) on the reference
modules, gradients from torch autograd.

Runs only in the build container (needs /root/reference); only its output fixture travels.
Import shims as in tools/gen_golden.py (timm.layers.Mlp restated, torchcfm placeholders).
The posterior's eps is torch.randn_like(mean) on the global CPU generator (dists.py:20): it is
captured by reseeding the generator before the reference call.
"""
import os
import sys

import torch
from safetensors.torch import save_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import gen_golden as G  # noqa: E402  (installs the import shims, loads the reference modules)
from oracle import vita_oracle as O  # noqa: E402

avae = G.load_by_path("ref_action_vae", "flare/networks/vita/action_vae.py")
F = torch.nn.functional


def ref_compute_loss_vae(flow, enc, dec, obs, FM, obs_features, gt, t, K, kl_weight, seed, training=True):
    """vita_policy.py:112-189 with use_action_vae, on the reference modules; torchcfm's draw restated with explicit t."""
    obs_lat = obs(obs_features)
    torch.manual_seed(seed)
    posterior, act_lat = enc(gt, deterministic=not training)                    # :121
    tt = t[:, None]
    xt = tt * act_lat + (1 - tt) * obs_lat
    ut = act_lat - obs_lat
    flow_loss = torch.mean((flow(xt, t) - ut) ** 2)
    loss = flow_loss
    m = dict(flow_loss=flow_loss)
    if kl_weight > 0:                                                           # :146-150
        kl = posterior.kl().mean()
        m["action_kl_loss"] = kl
        loss = loss + kl_weight * kl
    pred = FM.sample(flow, (obs_lat.shape[0], obs_lat.shape[1]), "cpu", start=obs_lat, num_steps=K)
    cons = F.mse_loss(pred, act_lat)
    r1 = F.l1_loss(dec(pred), gt)
    r2 = F.l1_loss(dec(act_lat), gt)
    loss = loss + 1.0 * cons + 0.5 * r1 + 0.5 * r2
    m.update(consistency_loss=cons, flow_action_recon_loss=r1, enc_action_recon_loss=r2, loss=loss)
    return loss, m, posterior, act_lat


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    A, D, H, L, R, T, aeh, ael, B, ted = 7, 32, 32, 2, 4, 16, 32, 4, 5, 16
    obs_dim = 32 + A
    tag = "vae_A7"
    out = {}
    sd = O.make_state_dict(obs_dim, A, D, H, L, R, T, aeh, ael, time_embed_dim=ted, encoder_type="cnn", seed=321, variational=True)
    flow, _, dec, obs = G.build_ref({k: v for k, v in sd.items() if not k.startswith("action_encoder.")} |
                                    {k: v for k, v in O.make_state_dict(obs_dim, A, D, H, L, R, T, aeh, ael, time_embed_dim=ted,
                                                                        encoder_type="cnn", seed=321).items()
                                     if k.startswith("action_encoder.")},
                                    obs_dim, A, D, H, L, R, T, aeh, ael, "cnn", ted)
    enc = avae.CNNActionEncoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=aeh, num_layers=ael)
    enc.load_state_dict(G.sub(sd, "action_encoder."), strict=True)
    FM = G.fmm.TorchFlowMatcher(fm=None, num_sampling_steps=6)
    g = torch.Generator().manual_seed(77)
    obs_features = torch.randn(B, obs_dim, generator=g)
    actions = torch.randn(B, T, A, generator=g)
    t = torch.rand(B, generator=g)
    for k, v in sd.items():
        out[f"{tag}/sd/{k}"] = v
    out[f"{tag}/obs_features"], out[f"{tag}/actions"], out[f"{tag}/t"] = obs_features, actions, t
    seed = 4242
    with torch.no_grad():
        torch.manual_seed(seed)
        posterior, z = enc(actions)
        torch.manual_seed(seed)
        eps = torch.randn_like(posterior.mean)
        out[f"{tag}/eps"] = eps
        out[f"{tag}/mean"], out[f"{tag}/logvar"], out[f"{tag}/z"] = posterior.mean.clone(), posterior.logvar.clone(), z
        om, ol, oz = O.vae_cnn_action_encode(sd, actions, eps)
        G.eq(om, posterior.mean, "vae mean"); G.eq(ol, posterior.logvar, "vae logvar"); G.eq(oz, z, "vae sample")
        pd, zd = enc(actions, deterministic=True)
        G.eq(O.vae_cnn_action_encode(sd, actions, None)[2], zd, "vae deterministic sample")
        assert float(pd.kl().abs().max()) == 0.0
    for name, klw, training in (("kl1e-4", 1e-4, True), ("kl0.1", 0.1, True), ("eval", 0.1, False)):
        for mod in (flow, enc, dec, obs):
            mod.zero_grad(set_to_none=True)
        of = obs_features.clone().requires_grad_(True)
        loss, met, _, _ = ref_compute_loss_vae(flow, enc, dec, obs, FM, of, actions, t, 6, klw, seed, training)
        loss.backward()
        oloss, omet = O.compute_loss(sd, obs_features, actions, t, 6, kl_weight=klw, enc_noise=eps if training else None)
        for k, v in met.items():
            out[f"{tag}/{name}/loss/{k}"] = v.detach().reshape(1)
            G.eq(omet[k].reshape(1), v.detach().reshape(1), f"vae {name} {k}")
        out[f"{tag}/{name}/grad/obs_features"] = of.grad
        for prefix, mod in (("flow_net.", flow), ("action_encoder.", enc), ("action_decoder.", dec), ("obs_encoder.", obs)):
            for n, p in mod.named_parameters():
                out[f"{tag}/{name}/grad/{prefix}{n}"] = p.grad.clone()
    path = os.path.join(REPO, "tests", "golden", "vae.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print("vae:", len(out), "tensors ->", path)


if __name__ == "__main__":
    main()
