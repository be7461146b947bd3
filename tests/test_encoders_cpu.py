"""vita_amd.encoders.TransformerActionEncoder (policy.action_ae.net.encoder_type: transformer) against the reference's own
module (action_ae.py:36-108), through tests/golden/transformer_encoder.safetensors (tools/gen_golden_transformer.py: output
and autograd gradients of the reference module on seeded weights; timm's Mlp is the restated stand-in there — UNPINNED).
The checkpoint keys are the reference's: load_state_dict(strict=True)."""
import os

import pytest
import torch
from safetensors import safe_open

from vita_amd.encoders import TransformerActionEncoder


@pytest.fixture(scope="module")
def gold(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "transformer_encoder.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_transformer_action_encoder_matches_the_reference(gold, tag):
    HID, D, NH, T, A, NL, B = gold[f"{tag}/dims"].tolist()
    enc = TransformerActionEncoder(enc_hidden_dim=HID, latent_dim=D, num_heads=NH, pred_horizon=T, action_dim=A, num_layers=NL)
    sd = {k[len(tag) + 4:]: v for k, v in gold.items() if k.startswith(f"{tag}/sd/")}
    enc.load_state_dict(sd, strict=True)          # the reference's parameter and buffer names
    actions = gold[f"{tag}/actions"].clone().requires_grad_(True)
    z = enc(actions)
    assert tuple(z.shape) == (B, D)
    assert (z - gold[f"{tag}/z"]).abs().max().item() <= 2e-6
    (z * gold[f"{tag}/w"]).sum().backward()
    assert (actions.grad - gold[f"{tag}/d_actions"]).abs().max().item() <= 2e-6
    for n, p in enc.named_parameters():
        ref = gold[f"{tag}/grad/{n}"]
        assert (p.grad - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()), n


def test_transformer_encoder_rejects_dropout():
    with pytest.raises(NotImplementedError):
        TransformerActionEncoder(64, 32, 8, 16, 7, dropout=0.1)


def test_variational_transformer_encoder_matches_the_reference(gold):
    """action_vae.py:37-120: the posterior parameters, their gradients, and the sample / KL formulas VitaPolicy.compute_loss
    applies to them (models/dists.py:4-35) against the reference's DiagonalGaussianDistribution on the same parameters."""
    tag = "vae"
    HID, D, NH, T, A, NL, B = gold[f"{tag}/dims"].tolist()
    enc = TransformerActionEncoder(enc_hidden_dim=HID, latent_dim=D, num_heads=NH, pred_horizon=T, action_dim=A, num_layers=NL, variational=True)
    enc.load_state_dict({k[len(tag) + 4:]: v for k, v in gold.items() if k.startswith(f"{tag}/sd/")}, strict=True)
    actions = gold[f"{tag}/actions"].clone().requires_grad_(True)
    params = enc(actions)
    assert tuple(params.shape) == (B, 2 * D)
    assert (params - gold[f"{tag}/params"]).abs().max().item() <= 2e-6
    (params * gold[f"{tag}/w"]).sum().backward()
    assert (actions.grad - gold[f"{tag}/d_actions"]).abs().max().item() <= 5e-6
    for n, p in enc.named_parameters():
        ref = gold[f"{tag}/grad/{n}"]
        assert (p.grad - ref).abs().max().item() <= 5e-6 * max(1.0, ref.abs().max().item()), n
    mean, logvar = torch.chunk(gold[f"{tag}/params"], 2, dim=1)
    z = mean + torch.exp(0.5 * logvar) * gold[f"{tag}/eps"]
    kl = 0.5 * torch.sum(torch.pow(mean, 2) + torch.exp(logvar) - 1.0 - logvar, dim=1)
    assert torch.equal(z, gold[f"{tag}/sample"]) and (kl - gold[f"{tag}/kl"]).abs().max().item() <= 1e-6 * gold[f"{tag}/kl"].abs().max().item()
