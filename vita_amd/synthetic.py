"""Deterministic synthetic weights with the reference's state-dict keys and shapes.

Shared by bench.py, the smoke test, the tests and (re-exported) the oracle.  The product path never needs it: it takes
whatever state dict the caller loads.  Lives in the package, not in oracle/, so that benchmark legs that run on the GPU
import nothing from the test infrastructure."""
from __future__ import annotations

import math
from typing import Dict

import torch

SD = Dict[str, torch.Tensor]


def make_state_dict(obs_dim: int, action_dim: int, latent_dim: int = 512, hidden_dim: int = 512,
                    num_layers: int = 4, mlp_ratio: int = 4, pred_horizon: int = 16,
                    ae_hidden: int = 512, ae_layers: int = 4, time_embed_dim: int = 256,
                    encoder_type: str = "cnn", seed: int = 0, dtype=torch.float32, variational: bool = False) -> SD:
    """Seeded Xavier-like weights with the reference's state-dict keys and shapes
    (SURVEY.md 8(b)).  Biases and the modulators are NOT zero so every term is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def lin(name, out_f, in_f, gain=1.0, bias_std=0.02):
        bound = gain * math.sqrt(6.0 / (in_f + out_f))
        sd[name + ".weight"] = ((torch.rand(out_f, in_f, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
        sd[name + ".bias"] = (torch.randn(out_f, generator=g, dtype=torch.float64) * bias_std).to(dtype)

    D, H, R = latent_dim, hidden_dim, mlp_ratio
    lin("obs_encoder", D, obs_dim)
    lin("flow_net.input_proj", H, D)
    lin("flow_net.time_embed.1", time_embed_dim * 4, time_embed_dim)
    lin("flow_net.time_embed.3", H, time_embed_dim * 4)
    for l in range(num_layers):
        lin(f"flow_net.layers.{l}.mlp.fc1", int(H * R), H)
        lin(f"flow_net.layers.{l}.mlp.fc2", H, int(H * R))
        lin(f"flow_net.layers.{l}.time_modulator.1", 3 * H, H, gain=0.5)
    sd["flow_net.norm.weight"] = (1 + 0.1 * torch.randn(H, generator=g, dtype=torch.float64)).to(dtype)
    sd["flow_net.norm.bias"] = (0.05 * torch.randn(H, generator=g, dtype=torch.float64)).to(dtype)
    lin("flow_net.out_proj", D, H)
    if encoder_type == "cnn":
        cin = action_dim
        for i in range(ae_layers):
            fan_in, fan_out = cin * 5, ae_hidden * 5
            bound = math.sqrt(6.0 / (fan_in + fan_out)) * 1.5
            sd[f"action_encoder.encoder.{2*i}.weight"] = (
                (torch.rand(ae_hidden, cin, 5, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
            sd[f"action_encoder.encoder.{2*i}.bias"] = (
                torch.randn(ae_hidden, generator=g, dtype=torch.float64) * 0.02).to(dtype)
            cin = ae_hidden
        conv_len = pred_horizon // (2 ** ae_layers)
        lin("action_encoder.latent_proj", 2 * D if variational else D, ae_hidden * conv_len)   # action_vae.py:213
    else:
        ald = D // pred_horizon
        lin("action_encoder.input_proj", ald, action_dim)
        for i in range(ae_layers):
            lin(f"action_encoder.layers.{i}.fc1", 4 * ald, ald)
            lin(f"action_encoder.layers.{i}.fc2", ald, 4 * ald)
    lin("action_decoder.input_proj", ae_hidden, D)
    for i in range(ae_layers):
        lin(f"action_decoder.layers.{i}.fc1", ae_hidden, ae_hidden)
        lin(f"action_decoder.layers.{i}.fc2", ae_hidden, ae_hidden)
    lin("action_decoder.output_proj", pred_horizon * action_dim, ae_hidden)
    return sd
