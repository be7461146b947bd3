"""Action encoders that run as ordinary torch modules beside the fused HIP path.

`VitaPolicy` hands the output of such a module to `vita_compute_loss_f32` as the action latents (VITA_ENC_EXTERNAL,
include/vita_hip.h) and gets d loss / d latents back, so autograd continues into the module: everything downstream of the
latents (flow matching, the K-step sampler and its backward, decoder, losses) stays one fused call.

`TransformerActionEncoder` is `policy.action_ae.net.encoder_type: transformer` (flare/networks/vita/action_ae.py:36-108 with
models/transformer/blocks.py:13-55, models/attention.py:24-66 and models/pos_embed.py:74-125): same constructor arguments,
same parameter / buffer names (so the reference's checkpoints load), same arithmetic order.  Its GEMMs, LayerNorms and the
17-token attention are torch-ROCm ops (rocBLAS / ATen), not hand-written kernels: the encoder is 4 % of a training step's
FLOPs and off the sampler path.  Dropout > 0 is not built (vita.yaml ships 0.0)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def create_sinusoidal_pos_embedding(num_positions: int, dimension: int) -> torch.Tensor:
    """action_ae.py:25-33: angle(pos, j) = pos / 10000^(2 (j // 2) / dimension) in float64, sin on even / cos on odd columns."""
    j = np.arange(dimension)
    table = np.arange(num_positions, dtype=np.float64)[:, None] / np.power(10000.0, 2 * (j // 2) / dimension)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.from_numpy(table).float()


class RotaryPosEmbed(nn.Module):
    """models/pos_embed.py:74-125 (torchtune's RoPE as the reference calls it): built over the FULL model dim, so the cache
    row of a position holds dim / 2 frequencies that are dealt to the heads in order (head h rotates with frequencies
    h * head_dim / 2 ... of theta) — kept as it is."""

    def __init__(self, dim: int, max_seq_len: int = 256, base: int = 10_000):
        super().__init__()
        self.dim, self.base, self.max_seq_len = dim, base, max_seq_len
        theta = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.register_buffer("theta", theta, persistent=False)
        seq_idx = torch.arange(max_seq_len, dtype=theta.dtype)
        idx_theta = torch.einsum("i, j -> ij", seq_idx, theta).float()
        self.register_buffer("cache", torch.stack([torch.cos(idx_theta), torch.sin(idx_theta)], dim=-1), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:      # (B, heads, S, head_dim)
        x = x.permute(0, 2, 1, 3)
        B, S, nh, hd = x.shape
        xs = x.float().reshape(B, S, nh, hd // 2, 2)
        rc = self.cache[:S].view(1, S, nh, hd // 2, 2)
        out = torch.stack([xs[..., 0] * rc[..., 0] - xs[..., 1] * rc[..., 1],
                           xs[..., 1] * rc[..., 0] + xs[..., 0] * rc[..., 1]], -1)
        return out.flatten(3).permute(0, 2, 1, 3).type_as(x)


class Attention(nn.Module):
    """models/attention.py:24-66: qkv Linear, LayerNorm over head_dim on q and k, RoPE on both, scaled dot-product attention."""

    def __init__(self, dim: int, num_heads: int = 8, max_seq_len: int = 16):
        super().__init__()
        if dim % num_heads:
            raise AssertionError("dim should be divisible by num_heads")
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.q_norm = nn.LayerNorm(self.head_dim)
        self.k_norm = nn.LayerNorm(self.head_dim)
        self.proj = nn.Linear(dim, dim)
        self.rope = RotaryPosEmbed(dim, max_seq_len=max_seq_len)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, S, C = x.shape
        qkv = self.qkv(x).reshape(B, S, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q, k = self.rope(self.q_norm(q)), self.rope(self.k_norm(k))
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj(x.transpose(1, 2).reshape(B, S, C))


class _Mlp(nn.Module):
    """timm.layers.Mlp (restated, UNPINNED like the other Mlp uses): fc1 -> act -> fc2, attribute names give the keys."""

    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU(approximate="tanh")      # blocks.py:10 approx_gelu
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class TransformerBlock(nn.Module):
    """models/transformer/blocks.py:13-55: pre-norm attention and MLP, each with a residual."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, rope_max_seq_length: int = 16):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.self_attn = Attention(dim, num_heads=num_heads, max_seq_len=rope_max_seq_length)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.LayerNorm(dim)

    def forward(self, x):
        x = x + self.self_attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class TransformerActionEncoder(nn.Module):
    """action_ae.py:36-108: a CLS token + one token per action step, fixed sinusoidal positions, `num_layers` blocks, LayerNorm,
    and the CLS output projected to the latent.  variational=True is the encoder of action_vae.py:37-120 (`use_variational`
    with `encoder_type: transformer`): the projection is 2 * latent_dim wide and forward returns those (B, 2 * latent_dim)
    posterior parameters [mean | logvar]; the caller (VitaPolicy.compute_loss) samples and forms the KL term (models/dists.py)."""

    def __init__(self, enc_hidden_dim: int, latent_dim: int, num_heads: int, pred_horizon: int, action_dim: int, num_layers: int = 4,
                 mlp_ratio: float = 4.0, dropout: float = 0.0, variational: bool = False):
        super().__init__()
        self.variational = bool(variational)
        if float(dropout) != 0.0:
            raise NotImplementedError("TransformerActionEncoder: dropout > 0 is not built (vita.yaml ships 0.0)")
        self.cls_embed = nn.Embedding(1, enc_hidden_dim)
        self.action_input_proj = nn.Linear(action_dim, enc_hidden_dim)
        self.layers = nn.ModuleList([TransformerBlock(enc_hidden_dim, num_heads, mlp_ratio, rope_max_seq_length=pred_horizon + 1)
                                     for _ in range(num_layers)])
        self.norm = nn.LayerNorm(enc_hidden_dim)
        self.latent_output_proj = nn.Linear(enc_hidden_dim, latent_dim * (2 if self.variational else 1))
        self.register_buffer("pos_embed", create_sinusoidal_pos_embedding(1 + pred_horizon, enc_hidden_dim).unsqueeze(0))
        for m in self.modules():        # action_ae.py:80-85
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, actions: torch.Tensor) -> torch.Tensor:      # (B, T, A) -> (B, latent_dim), or (B, 2 * latent_dim) when variational
        B = actions.shape[0]
        tokens = torch.cat([self.cls_embed.weight.unsqueeze(0).expand(B, -1, -1), self.action_input_proj(actions)], dim=1)
        x = tokens + self.pos_embed[:, :tokens.shape[1]]
        for layer in self.layers:
            x = layer(x)
        return self.latent_output_proj(self.norm(x)[:, 0])
