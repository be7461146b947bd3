#!/usr/bin/env python3
"""tests/golden/transformer_encoder.safetensors: the reference's TransformerActionEncoder (action_ae.py:36-108) on seeded
weights — state dict, input, output and autograd gradients (build container only; only the fixture travels).
Uses the loaders / import shims of tools/gen_golden.py (timm.layers.Mlp is the restated stand-in: UNPINNED there)."""
import os
import sys

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd.encoders import TransformerActionEncoder  # noqa: E402


def main():
    torch.set_num_threads(1)
    out = {}
    for tag, (HID, D, NH, T, A, NL, B) in {"small": (64, 32, 8, 16, 7, 2, 5), "wide": (64, 48, 4, 8, 3, 1, 3)}.items():
        torch.manual_seed(7)
        ref = G.aae.TransformerActionEncoder(enc_hidden_dim=HID, latent_dim=D, num_heads=NH, pred_horizon=T, action_dim=A,
                                             num_layers=NL, mlp_ratio=4.0, dropout=0.0)
        with torch.no_grad():       # make biases / norms non-trivial (the reference initialises them to 0 / 1)
            g = torch.Generator().manual_seed(3)
            for n, p in ref.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        ref.train()
        actions = torch.randn(B, T, A, generator=torch.Generator().manual_seed(5), requires_grad=True)
        w = torch.randn(B, D, generator=torch.Generator().manual_seed(6))
        z = ref(actions)
        (z * w).sum().backward()
        sd = ref.state_dict()
        mine = TransformerActionEncoder(enc_hidden_dim=HID, latent_dim=D, num_heads=NH, pred_horizon=T, action_dim=A, num_layers=NL)
        missing = mine.load_state_dict(sd, strict=True)
        a2 = actions.detach().clone().requires_grad_(True)
        z2 = mine(a2)
        (z2 * w).sum().backward()
        assert (z2 - z).abs().max().item() <= 1e-6, (z2 - z).abs().max().item()
        assert (a2.grad - actions.grad).abs().max().item() <= 1e-6
        for (n, p), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
            assert n == n2 and (p.grad - p2.grad).abs().max().item() <= 2e-6 * max(1.0, p.grad.abs().max().item()), n
        for k, v in sd.items():
            out[f"{tag}/sd/{k}"] = v.contiguous()
        out[f"{tag}/actions"], out[f"{tag}/w"], out[f"{tag}/z"], out[f"{tag}/d_actions"] = actions.detach(), w, z.detach(), actions.grad
        for n, p in ref.named_parameters():
            out[f"{tag}/grad/{n}"] = p.grad.contiguous()
        out[f"{tag}/dims"] = torch.tensor([HID, D, NH, T, A, NL, B])
    # the variational encoder of action_vae.py:37-120 (use_variational + encoder_type transformer): posterior parameters
    avae = G.load_by_path("ref_action_vae", "flare/networks/vita/action_vae.py")
    HID, D, NH, T, A, NL, B = 64, 32, 8, 16, 7, 2, 5
    torch.manual_seed(9)
    ref = avae.TransformerActionEncoder(enc_hidden_dim=HID, latent_dim=D, num_heads=NH, pred_horizon=T, action_dim=A, num_layers=NL,
                                        mlp_ratio=4.0, dropout=0.0)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    ref.train()
    actions = torch.randn(B, T, A, generator=torch.Generator().manual_seed(5), requires_grad=True)
    w = torch.randn(B, 2 * D, generator=torch.Generator().manual_seed(6))
    post, z = ref(actions, deterministic=True)
    assert torch.equal(z, post.mean) and torch.equal(post.parameters[:, :D], post.mean)
    (post.parameters * w).sum().backward()
    mine = TransformerActionEncoder(enc_hidden_dim=HID, latent_dim=D, num_heads=NH, pred_horizon=T, action_dim=A, num_layers=NL, variational=True)
    mine.load_state_dict(ref.state_dict(), strict=True)
    a2 = actions.detach().clone().requires_grad_(True)
    p2 = mine(a2)
    (p2 * w).sum().backward()
    assert (p2 - post.parameters).abs().max().item() <= 1e-6 and (a2.grad - actions.grad).abs().max().item() <= 1e-6
    # the posterior's sample / KL (models/dists.py) on these parameters, for the policy-side restatement
    eps = torch.randn(B, D, generator=torch.Generator().manual_seed(8))
    torch.manual_seed(123)
    post_s = G.load_by_path("ref_dists", "flare/models/dists.py").DiagonalGaussianDistribution(post.parameters.detach())
    for k, v in ref.state_dict().items():
        out[f"vae/sd/{k}"] = v.contiguous()
    out["vae/actions"], out["vae/w"], out["vae/params"], out["vae/d_actions"] = actions.detach(), w, post.parameters.detach(), actions.grad
    out["vae/eps"], out["vae/sample"], out["vae/kl"] = eps, (post_s.mean + post_s.std * eps), post_s.kl()
    for n, p in ref.named_parameters():
        out[f"vae/grad/{n}"] = p.grad.contiguous()
    out["vae/dims"] = torch.tensor([HID, D, NH, T, A, NL, B])
    save_file(out, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "transformer_encoder.safetensors"))
    print("ok", len(out), "tensors")


if __name__ == "__main__":
    main()
