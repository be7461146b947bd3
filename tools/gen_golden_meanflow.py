#!/usr/bin/env python3
"""tests/golden/meanflow_train.safetensors: VitaPolicy.compute_loss with the MeanFlow matcher replayed on the REFERENCE's
own modules — flare/flow/mean_flow_matcher.py MeanFlowMatcher.compute_loss / .sample, flare/policies/vita/
simple_mean_flow_net.py SimpleMeanFlowNet, action_ae.py CNNActionEncoder / SimpleActionDecoder — with gradients from torch
autograd.  The (t, r) draw of sample_t_r is fixed by overriding that one method on the matcher instance (the randomness is
the caller's in the build; everything downstream of the draw is the reference's code).  Runs only in the build container."""
import os
import sys

import torch
import torch.nn as nn
from safetensors.torch import save_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import gen_golden as G  # noqa: E402
from oracle import vita_oracle as O  # noqa: E402

smf = G.load_by_path("ref_simple_mean_flow_net", "flare/policies/vita/simple_mean_flow_net.py")
F = torch.nn.functional


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    A, D, H, L, T, aeh, ael, B = 7, 32, 48, 2, 16, 32, 4, 6
    obs_dim = 32 + A
    out = {}
    base = O.make_state_dict(obs_dim, A, D, 32, 1, 4, T, aeh, ael, time_embed_dim=16, encoder_type="cnn", seed=77)
    sd = {k: v for k, v in base.items() if not k.startswith("flow_net.")}
    sd.update(O.make_meanflow_state_dict(D, H, L, 4.0, seed=78))
    net = smf.SimpleMeanFlowNet(D, H, D, L, mlp_ratio=4.0, dropout=0.0)
    net.load_state_dict(G.sub(sd, "flow_net."), strict=True)
    enc = G.aae.CNNActionEncoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=aeh, num_layers=ael)
    enc.load_state_dict(G.sub(sd, "action_encoder."), strict=True)
    dec = G.aae.SimpleActionDecoder(dec_hidden_dim=aeh, latent_dim=D, pred_horizon=T, action_dim=A, num_layers=ael)
    dec.load_state_dict(G.sub(sd, "action_decoder."), strict=True)
    obs = nn.Linear(obs_dim, D)
    obs.load_state_dict(G.sub(sd, "obs_encoder."), strict=True)
    g = torch.Generator().manual_seed(79)
    obs_features = torch.randn(B, obs_dim, generator=g)
    actions = torch.randn(B, T, A, generator=g)
    # a (t, r) pair as sample_t_r makes them: t >= r in (0, 1), r == t on some rows (flow_ratio)
    smp = torch.sigmoid(torch.randn(B, 2, generator=g) - 0.4)
    t, r = smp.max(1)[0], smp.min(1)[0]
    r[:2] = t[:2]
    for k, v in sd.items():
        out[f"mf/sd/{k}"] = v
    out["mf/obs_features"], out["mf/actions"], out["mf/t"], out["mf/r"] = obs_features, actions, t, r
    # (num_steps, use_imf, dispersive_loss_weight, dispersive_loss_tau)
    for K, use_imf, dw, tau in ((1, False, 0.0, 1.0), (3, False, 0.0, 1.0), (1, True, 0.0, 1.0), (2, True, 0.0, 1.0),
                                (1, False, 0.5, 1.0), (2, True, 0.25, 0.7)):
        tag = f"mf/{'imf_' if use_imf else ''}{'disp_' if dw > 0 else ''}K{K}"
        FM = (G.fmm.ImprovedMeanFlowConditionalFlowMatcher if use_imf else G.fmm.MeanFlowConditionalFlowMatcher)(
            num_sampling_steps=K, dispersive_loss_weight=dw, dispersive_loss_tau=tau)
        FM.sample_t_r = lambda batch_size, device: (t.clone(), r.clone())
        for mod in (net, enc, dec, obs):
            mod.zero_grad(set_to_none=True)
        of = obs_features.clone()
        # vita_policy.py:112-189 on the reference modules
        obs_lat = obs(of)
        act_lat = enc(actions)
        flow_loss, met = FM.compute_loss(net, target=act_lat, start=obs_lat)
        loss = flow_loss
        pred = FM.sample(net, shape=(B, D), device="cpu", start=obs_lat, num_steps=K)
        cons = F.mse_loss(pred, act_lat)
        r1 = F.l1_loss(dec(pred), actions)
        r2 = F.l1_loss(dec(act_lat), actions)
        loss = loss + 1.0 * cons + 0.5 * r1 + 0.5 * r2
        loss.backward()
        ref = dict(flow_loss=flow_loss, consistency_loss=cons, flow_action_recon_loss=r1, enc_action_recon_loss=r2, loss=loss)
        if not use_imf and dw == 0:
            ref["meanflow_loss"] = flow_loss
            assert abs(met["meanflow_loss"] - flow_loss.item()) < 1e-12
        oloss, omet = O.compute_loss_meanflow(sd, obs_features, actions, t, r, K, use_imf=use_imf, dispersive_loss_weight=dw,
                                              dispersive_loss_tau=tau)
        if dw > 0:
            out[f"{tag}/loss/dispersive_loss"] = torch.tensor([met["dispersive_loss"]], dtype=torch.float32)
            assert abs(omet["dispersive_loss"].item() - met["dispersive_loss"]) <= 1e-7 * abs(met["dispersive_loss"])
            if not use_imf:
                out[f"{tag}/loss/meanflow_loss"] = torch.tensor([met["meanflow_loss"]], dtype=torch.float32)
                assert abs(omet["meanflow_loss"].item() - met["meanflow_loss"]) <= 1e-7 * abs(met["meanflow_loss"])
        for k, v in ref.items():
            out[f"{tag}/loss/{k}"] = v.detach().reshape(1).clone()
            G.eq(omet[k].reshape(1), v.detach().reshape(1), f"meanflow {tag} {k}")
        if use_imf:     # the matcher reports these four as python floats (.item())
            for k in ("imf_loss", "aux_v_loss", "imf_mse", "aux_v_mse"):
                out[f"{tag}/loss/{k}"] = torch.tensor([met[k]], dtype=torch.float32)
                assert abs(omet[k].item() - met[k]) <= 1e-12 + 1e-7 * abs(met[k]), (k, omet[k].item(), met[k])
        for prefix, mod in (("flow_net.", net), ("action_encoder.", enc), ("action_decoder.", dec), ("obs_encoder.", obs)):
            for n, p in mod.named_parameters():
                out[f"{tag}/grad/{prefix}{n}"] = p.grad.clone() if p.grad is not None else torch.zeros_like(p)
    path = os.path.join(REPO, "tests", "golden", "meanflow_train.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print("meanflow_train:", len(out), "tensors ->", path)


if __name__ == "__main__":
    main()
