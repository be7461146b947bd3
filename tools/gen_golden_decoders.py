#!/usr/bin/env python3
"""tests/golden/cnn_decoders.safetensors: the reference's CNNActionDecoder and CNNAttentionActionDecoder
(action_vae.py:232-387) on seeded weights — state dict, input latents, output actions and autograd gradients (build container
only; only the fixture travels).  Uses the loaders / import shims of tools/gen_golden.py (timm.layers.Mlp is the restated
stand-in: UNPINNED there)."""
import os
import sys

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd.decoders import CNNActionDecoder, CNNAttentionActionDecoder  # noqa: E402

avae = G.load_by_path("ref_action_vae", "flare/networks/vita/action_vae.py")


def run(tag, ref, mine, D, B, out):
    with torch.no_grad():       # biases / norms non-trivial (the reference initialises them to 0 / 1)
        g = torch.Generator().manual_seed(3)
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    ref.train()
    z = torch.randn(B, D, generator=torch.Generator().manual_seed(5), requires_grad=True)
    a = ref(z)
    w = torch.randn(a.shape, generator=torch.Generator().manual_seed(6))
    (a * w).sum().backward()
    sd = ref.state_dict()
    mine.load_state_dict(sd, strict=True)
    z2 = z.detach().clone().requires_grad_(True)
    a2 = mine(z2)
    (a2 * w).sum().backward()
    assert a2.shape == a.shape and (a2 - a).abs().max().item() <= 1e-6, (a2 - a).abs().max().item()
    assert (z2.grad - z.grad).abs().max().item() <= 1e-6 * max(1.0, z.grad.abs().max().item())
    for (n, p), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
        assert n == n2 and (p.grad - p2.grad).abs().max().item() <= 2e-6 * max(1.0, p.grad.abs().max().item()), n
    for k, v in sd.items():
        out[f"{tag}/sd/{k}"] = v.contiguous()
    out[f"{tag}/z"], out[f"{tag}/w"], out[f"{tag}/actions"], out[f"{tag}/d_z"] = z.detach(), w, a.detach().contiguous(), z.grad
    for n, p in ref.named_parameters():
        out[f"{tag}/grad/{n}"] = p.grad.contiguous()


def main():
    torch.set_num_threads(1)
    out = {}
    # CNNActionDecoder: (pred_horizon, action_dim, latent_dim, hidden_dim, num_layers, B); "interp": 12 // 8 = 1 -> length 8,
    # linearly interpolated to 12 (action_vae.py:293-296)
    for tag, (T, A, D, HID, NL, B) in {"cnn_small": (16, 7, 32, 24, 3, 5), "cnn_full_len": (16, 3, 48, 16, 4, 3),
                                       "cnn_interp": (12, 4, 32, 16, 3, 4)}.items():
        torch.manual_seed(7)
        ref = avae.CNNActionDecoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=HID, num_layers=NL)
        mine = CNNActionDecoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=HID, num_layers=NL)
        run(tag, ref, mine, D, B, out)
        out[f"{tag}/dims"] = torch.tensor([T, A, D, HID, NL, B])
    # CNNAttentionActionDecoder: (pred_horizon, action_dim, latent_dim, hidden_dim, num_cnn_layers, num_transformer_layers, heads, B)
    for tag, (T, A, D, HID, NC, NT, NH, B) in {"attn_small": (16, 7, 32, 32, 2, 2, 4, 5), "attn_wide": (8, 3, 48, 64, 3, 1, 8, 3)}.items():
        torch.manual_seed(11)
        ref = avae.CNNAttentionActionDecoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=HID, num_cnn_layers=NC,
                                             num_transformer_layers=NT, num_heads=NH, mlp_ratio=4.0, dropout=0.0)
        mine = CNNAttentionActionDecoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=HID, num_cnn_layers=NC,
                                         num_transformer_layers=NT, num_heads=NH, mlp_ratio=4.0)
        run(tag, ref, mine, D, B, out)
        out[f"{tag}/dims"] = torch.tensor([T, A, D, HID, NC, NT, NH, B])
    save_file(out, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cnn_decoders.safetensors"))
    print("ok", len(out), "tensors", sum(v.numel() for v in out.values()) * 4 / 1e6, "MB")


if __name__ == "__main__":
    main()
