#!/usr/bin/env python3
"""tests/golden/alt.safetensors: the reference's SimpleMeanFlowNet, MeanFlowMatcher.sample and
ConsistencyFlowMatcher.sample run on seeded weights (build container only; only the fixture travels).
Uses the loaders / import shims of tools/gen_golden.py."""
import os
import sys

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (executes the shims, loads the reference files by path)

O = G.O
smf = G.load_by_path("ref_simple_mean_flow_net", "flare/policies/vita/simple_mean_flow_net.py")


def eq(a, b, what):
    assert torch.equal(a, b), f"oracle != reference at {what}: max |d| = {(a - b).abs().max().item():g}"


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}
    g = torch.Generator().manual_seed(11)
    # ---- MeanFlow: D = 32, H = 48, L = 3, ratio 4 -----------------------------------------------------------------
    D, H, L, B = 32, 48, 3, 7
    sd = O.make_meanflow_state_dict(D, H, L, 4.0, seed=3)
    net = smf.SimpleMeanFlowNet(D, H, D, L, mlp_ratio=4.0, dropout=0.0)
    net.load_state_dict(G.sub(sd, "flow_net."), strict=True)
    net.eval()
    x = torch.randn(B, D, generator=g)
    t = torch.rand(B, generator=g)
    h = torch.rand(B, generator=g) * t
    with torch.no_grad():
        u, v, feats = net(x, t, h)
    ou, ov, of = O.mean_flow_net_forward(sd, x, t, h)
    eq(ou, u, "meanflow u"); eq(ov, v, "meanflow v"); eq(torch.stack(of, 0), feats, "meanflow features")
    for k, val in sd.items():
        out["mf/sd/" + k] = val
    out["mf/x"], out["mf/t"], out["mf/h"], out["mf/u"], out["mf/v"] = x, t, h, u, v
    FM = G.fmm.MeanFlowConditionalFlowMatcher(num_sampling_steps=1)
    for K in (1, 2, 5):
        with torch.no_grad():
            xs = FM.sample(net, (B, D), torch.device("cpu"), num_steps=K, start=x)
        eq(O.mean_flow_sample(sd, x, K), xs, f"meanflow sample K={K}")
        out[f"mf/sample/K{K}"] = xs
    # ---- Consistency-FM sampler over SimpleFlowNet (tiny dims of the main fixtures) ----------------------------------
    D, H, L, B = 32, 32, 2, 5
    sd = O.make_state_dict(20, 7, latent_dim=D, hidden_dim=H, num_layers=L, mlp_ratio=4, pred_horizon=16, ae_hidden=32,
                           ae_layers=2, time_embed_dim=64, seed=4)
    flow = G.sfn.SimpleFlowNet(D, H, D, L, mlp_ratio=4, dropout=0.0, time_embed_dim=64)
    flow.load_state_dict(G.sub(sd, "flow_net."), strict=True)
    flow.eval()
    for k, val in sd.items():
        if k.startswith("flow_net."):
            out["cfm/sd/" + k] = val
    x0 = torch.randn(B, D, generator=g)
    out["cfm/x0"] = x0
    for K in (1, 4):
        # deterministic branch (:98)
        FMc = G.fmm.ConsistencyFlowMatcher(num_sampling_steps=K, sigma_var=0.0)
        with torch.no_grad():
            z = FMc.sample(flow, (B, D), torch.device("cpu"), start=x0)
        eq(O.consistency_sample(sd, x0, K, sigma_var=0.0), z, f"consistency det K={K}")
        out[f"cfm/det/K{K}"] = z
        # default sigma_var = 1.0 (:93-96): the reference draws torch.randn_like(pred_sigma) from the global CPU generator,
        # one draw per step; the same seed reproduces the draws for the fixture
        FMs = G.fmm.ConsistencyFlowMatcher(num_sampling_steps=K)
        torch.manual_seed(1234 + K)
        with torch.no_grad():
            z = FMs.sample(flow, (B, D), torch.device("cpu"), start=x0)
        torch.manual_seed(1234 + K)
        noise = torch.stack([torch.randn(B, D) for _ in range(K)], 0)
        eq(O.consistency_sample(sd, x0, K, noise=noise), z, f"consistency sde K={K}")
        out[f"cfm/sde/K{K}"], out[f"cfm/sde/noise{K}"] = z, noise
    path = os.path.join(G.REPO, "tests", "golden", "alt.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print("alt:", len(out), "tensors ->", path)


if __name__ == "__main__":
    main()
