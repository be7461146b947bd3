"""Action decoders that run as ordinary torch modules beside the fused HIP path.

`VitaPolicy` runs the fused step in two calls around such a module (vita_loss_config.external_decoder, include/vita_hip.h):
the forward call hands out the latents [pred; z1], the module decodes them and forms the reconstruction terms under torch
autograd, and the backward call takes d (those terms) / d latents and unrolls the sampler, the flow-matching term and the
encoders from there.  At inference the module decodes what the (one-launch) sampler returns.

`CNNActionDecoder` is `policy.action_ae.net.decoder_type: cnn` of the variational autoencoder (flare/networks/vita/
action_vae.py:232-298, built at :422-429); `CNNAttentionActionDecoder` is action_vae.py:301-387 (defined by the reference but
reached by none of its factories; buildable here by passing it as `VitaPolicy(action_decoder=...)`).  Same constructor
arguments, same parameter names (the reference's checkpoints load), same arithmetic order.  Their transposed convolutions and
the 16-token attention are torch-ROCm ops (MIOpen / rocBLAS / ATen), not hand-written kernels: off the default path
(vita.yaml: decoder_type simple) and off the sampler.  Dropout > 0 is not built."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoders import TransformerBlock


def _upsampling_stack(hidden_dim: int, out_dim: int, num_layers: int, relu_after_last: bool) -> nn.Sequential:
    """ConvTranspose1d(k5, s2, p2, output_padding 1) doubles the length; ReLU between layers (action_vae.py:254-268, 327-335:
    the Sequential's indices — conv at 2i, ReLU at 2i + 1 — are the state-dict keys)."""
    mods = []
    for i in range(num_layers):
        last = i == num_layers - 1
        mods.append(nn.ConvTranspose1d(hidden_dim, out_dim if last else hidden_dim, kernel_size=5, stride=2, padding=2, output_padding=1))
        if not last or relu_after_last:
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class CNNActionDecoder(nn.Module):
    """action_vae.py:232-298."""

    def __init__(self, pred_horizon: int, action_dim: int, latent_dim: int, hidden_dim: int = 128, num_layers: int = 3):
        super().__init__()
        self.pred_horizon, self.action_dim, self.latent_dim = pred_horizon, action_dim, latent_dim
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.conv_output_length = pred_horizon // (2 ** num_layers)
        if self.conv_output_length < 1:
            raise ValueError(f"CNNActionDecoder: pred_horizon {pred_horizon} < 2^num_layers = {2 ** num_layers}")
        self.latent_proj = nn.Linear(latent_dim, hidden_dim * self.conv_output_length)
        self.decoder = _upsampling_stack(hidden_dim, action_dim, num_layers, relu_after_last=False)
        for m in self.modules():        # :274-282
            if isinstance(m, (nn.Linear, nn.ConvTranspose1d)):
                nn.init.orthogonal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.fill_(0.0)

    def forward(self, z: torch.Tensor) -> torch.Tensor:      # (B, latent_dim) -> (B, pred_horizon, action_dim)
        x = self.latent_proj(z).view(z.shape[0], self.hidden_dim, self.conv_output_length)
        x = self.decoder(x)
        if x.shape[2] != self.pred_horizon:         # :293-296 (pred_horizon not a multiple of 2^num_layers)
            x = F.interpolate(x, size=self.pred_horizon, mode="linear", align_corners=False)
        return x.transpose(1, 2)


class CNNAttentionActionDecoder(nn.Module):
    """action_vae.py:301-387: transposed-convolution upsampling to one token per action step, `num_transformer_layers` pre-norm
    blocks (RoPE attention, models/transformer/blocks.py:13-55), LayerNorm, a per-token action head."""

    def __init__(self, pred_horizon: int, action_dim: int, latent_dim: int, hidden_dim: int = 128, num_cnn_layers: int = 2,
                 num_transformer_layers: int = 2, num_heads: int = 4, mlp_ratio: float = 4.0, dropout: float = 0.0):
        super().__init__()
        if float(dropout) != 0.0:
            raise NotImplementedError("CNNAttentionActionDecoder: dropout > 0 is not built")
        self.pred_horizon, self.action_dim, self.latent_dim, self.hidden_dim = pred_horizon, action_dim, latent_dim, hidden_dim
        self.conv_output_length = pred_horizon // (2 ** num_cnn_layers)
        if self.conv_output_length < 1:
            raise ValueError(f"CNNAttentionActionDecoder: pred_horizon {pred_horizon} < 2^num_cnn_layers = {2 ** num_cnn_layers}")
        self.latent_proj = nn.Linear(latent_dim, hidden_dim * self.conv_output_length)
        self.cnn_decoder = _upsampling_stack(hidden_dim, hidden_dim, num_cnn_layers, relu_after_last=True)
        self.token_proj = nn.Linear(hidden_dim, hidden_dim)
        self.transformer_layers = nn.ModuleList([TransformerBlock(hidden_dim, num_heads, mlp_ratio, rope_max_seq_length=pred_horizon)
                                                 for _ in range(num_transformer_layers)])
        self.norm = nn.LayerNorm(hidden_dim)
        self.action_head = nn.Linear(hidden_dim, action_dim)
        for m in self.modules():        # :359-367
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.ConvTranspose1d):
                nn.init.orthogonal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.fill_(0.0)

    def forward(self, z: torch.Tensor) -> torch.Tensor:      # (B, latent_dim) -> (B, pred_horizon, action_dim)
        x = self.latent_proj(z).view(z.shape[0], self.hidden_dim, self.conv_output_length)
        x = self.token_proj(self.cnn_decoder(x).transpose(1, 2))
        for layer in self.transformer_layers:
            x = layer(x)
        return self.action_head(self.norm(x))
