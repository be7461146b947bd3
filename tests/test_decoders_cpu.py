"""vita_amd.decoders (policy.action_ae.net.decoder_type: cnn, and the attention variant) against the reference's own modules
(action_vae.py:232-387), through tests/golden/cnn_decoders.safetensors (tools/gen_golden_decoders.py: output and autograd
gradients of the reference modules on seeded weights; timm's Mlp inside the transformer blocks is the restated stand-in —
UNPINNED).  The checkpoint keys are the reference's: load_state_dict(strict=True).  The oracle's restatement of the CNN decoder
(what the GPU tests of the external-decoder step check against) is pinned by the same vectors."""
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vita_oracle as O
from vita_amd.decoders import CNNActionDecoder, CNNAttentionActionDecoder


@pytest.fixture(scope="module")
def gold(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "cnn_decoders.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


def _check(gold, tag, dec, B, T, A):
    sd = {k[len(tag) + 4:]: v for k, v in gold.items() if k.startswith(f"{tag}/sd/")}
    dec.load_state_dict(sd, strict=True)          # the reference's parameter names
    z = gold[f"{tag}/z"].clone().requires_grad_(True)
    a = dec(z)
    assert tuple(a.shape) == (B, T, A)
    assert (a - gold[f"{tag}/actions"]).abs().max().item() <= 2e-6
    (a * gold[f"{tag}/w"]).sum().backward()
    assert (z.grad - gold[f"{tag}/d_z"]).abs().max().item() <= 2e-6 * max(1.0, gold[f"{tag}/d_z"].abs().max().item())
    for n, p in dec.named_parameters():
        ref = gold[f"{tag}/grad/{n}"]
        # 5e-6: fp32 summation order differs with the thread count (the fixture was made with one thread)
        assert (p.grad - ref).abs().max().item() <= 5e-6 * max(1.0, ref.abs().max().item()), n
    return sd


@pytest.mark.parametrize("tag", ["cnn_small", "cnn_full_len", "cnn_interp"])
def test_cnn_action_decoder_matches_the_reference(gold, tag):
    T, A, D, HID, NL, B = gold[f"{tag}/dims"].tolist()
    sd = _check(gold, tag, CNNActionDecoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=HID, num_layers=NL), B, T, A)
    # the oracle's functional restatement, in fp64
    sd64 = {"action_decoder." + k: v.double() for k, v in sd.items()}
    a = O.cnn_action_decode(sd64, gold[f"{tag}/z"].double(), T, HID)
    assert (a - gold[f"{tag}/actions"].double()).abs().max().item() <= 2e-6


@pytest.mark.parametrize("tag", ["attn_small", "attn_wide"])
def test_cnn_attention_action_decoder_matches_the_reference(gold, tag):
    T, A, D, HID, NC, NT, NH, B = gold[f"{tag}/dims"].tolist()
    _check(gold, tag, CNNAttentionActionDecoder(pred_horizon=T, action_dim=A, latent_dim=D, hidden_dim=HID, num_cnn_layers=NC,
                                                num_transformer_layers=NT, num_heads=NH), B, T, A)


def test_decoders_reject_what_is_not_built():
    with pytest.raises(NotImplementedError):
        CNNAttentionActionDecoder(16, 7, 32, dropout=0.1)
    with pytest.raises(ValueError):
        CNNActionDecoder(pred_horizon=4, action_dim=2, latent_dim=8, hidden_dim=8, num_layers=3)
