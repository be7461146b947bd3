"""CPU oracle for the VITA vision-latent -> action-latent flow-matching hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path (vita_amd/) never
imports it and fails loudly when the HIP library is missing.

It is a functional (no nn.Module) restatement, on torch CPU tensors, of the reference
algorithm, op for op, each function citing the reference file:line it follows.  Weights
come in as a flat ``dict[str, Tensor]`` keyed by the reference's state-dict names
(SURVEY.md section 8(b)), so a reference checkpoint drops in unchanged.

Pinning (tools/gen_golden.py, run in the build container where /root/reference exists):
  * PINNED against the reference's own modules loaded from /root/reference by file path:
    SinusoidalPosEmbed, SimpleFlowNet (+FlowNetLayer), TorchFlowMatcher.sample,
    CNNActionEncoder, SimpleActionEncoder, SimpleActionDecoder, Normalize/Unnormalize,
    compute_contrastive_loss, DiagonalGaussianDistribution.kl.  Outputs are committed
    under tests/golden/ and test_oracle_golden.py checks this file against them.
  * RESTATED, UNPINNED (third-party code absent from this image and from /root/reference):
    timm==1.0.20 ``timm.layers.Mlp`` op order (fc1 -> act -> drop -> norm(Identity) ->
    fc2 -> drop); torchcfm==1.0.5 ``ConditionalFlowMatcher.sample_location_and_conditional_flow``
    (xt = t*x1 + (1-t)*x0 + sigma*eps, ut = x1 - x0) and the exact-OT re-pairing (POT);
    diffusers==0.30.3 EMAModel decay and cosine-with-warmup schedule.  These follow the
    published sources of the pinned versions; none of the reference's files hold outputs
    for them.

All functions are dtype-generic: pass float64 weights/inputs to get the fp64 "truth"
used to bound the fp32 error of both the reference and the HIP path.
"""
from __future__ import annotations

import math
from typing import Callable

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = dict  # state dict: reference key -> tensor


# ----------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------
def sinusoidal_pos_embed(t: Tensor, dim: int = 256) -> Tensor:
    """flare/models/pos_embed.py:11-18.  t (B,) -> (B, dim) = cat(sin(t f), cos(t f))."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    # reference builds the frequency table in the default dtype (fp32) from arange
    freqs = torch.exp(torch.arange(half, device=t.device) * -emb)
    if t.dtype == torch.float64:
        freqs = torch.exp(torch.arange(half, device=t.device, dtype=torch.float64) * -emb)
    e = t[:, None] * freqs[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def timm_mlp(x: Tensor, sd: SD, prefix: str, act: Callable[[Tensor], Tensor]) -> Tensor:
    """timm==1.0.20 timm.layers.Mlp.forward (RESTATED, UNPINNED): fc2(act(fc1(x))) with
    drop=0 and norm=Identity.  Call sites: simple_flow_net.py:19-24, action_ae.py:139-146,
    245-252."""
    h = F.linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"])
    h = act(h)
    return F.linear(h, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"])


def gelu_tanh(x: Tensor) -> Tensor:
    return F.gelu(x, approximate="tanh")


def gelu_exact(x: Tensor) -> Tensor:
    return F.gelu(x)


# ----------------------------------------------------------------------------------------
# SimpleFlowNet  (flare/policies/vita/simple_flow_net.py)
# ----------------------------------------------------------------------------------------
def flow_net_num_layers(sd: SD, prefix: str = "flow_net.") -> int:
    n = 0
    while f"{prefix}layers.{n}.mlp.fc1.weight" in sd:
        n += 1
    return n


def flow_net_time_embed(sd: SD, t: Tensor, prefix: str = "flow_net.") -> Tensor:
    """simple_flow_net.py:66-71,101: Linear(1024->H) o Mish o Linear(256->1024) o Sinusoid(256)."""
    time_embed_dim = sd[prefix + "time_embed.1.weight"].shape[1]
    e = sinusoidal_pos_embed(t, time_embed_dim)
    e = F.linear(e, sd[prefix + "time_embed.1.weight"], sd[prefix + "time_embed.1.bias"])
    e = F.mish(e)
    return F.linear(e, sd[prefix + "time_embed.3.weight"], sd[prefix + "time_embed.3.bias"])


def flow_net_modulation(sd: SD, te: Tensor, layer: int, prefix: str = "flow_net."):
    """simple_flow_net.py:27-30,41-42: Linear(H->3H)(SiLU(te)).view(B,3,H) -> gamma, scale, shift."""
    p = f"{prefix}layers.{layer}.time_modulator.1."
    m = F.linear(F.silu(te), sd[p + "weight"], sd[p + "bias"])
    B = te.shape[0]
    H = m.shape[1] // 3
    gamma, scale, shift = m.view(B, 3, H).unbind(1)
    return gamma, scale, shift


def flow_net_layer(sd: SD, x: Tensor, te: Tensor, layer: int, prefix: str = "flow_net.") -> Tensor:
    """FlowNetLayer.forward, simple_flow_net.py:39-48."""
    gamma, scale, shift = flow_net_modulation(sd, te, layer, prefix)
    H = x.shape[-1]
    xn = F.layer_norm(x, (H,), None, None, 1e-6)           # :18 LayerNorm(no affine, eps=1e-6)
    xn = xn.mul(scale.add(1)).add(shift)                   # :45
    y = timm_mlp(xn, sd, f"{prefix}layers.{layer}.mlp.", gelu_tanh)
    return x + y.mul(gamma)                                # :46


def flow_net_forward(sd: SD, x: Tensor, t: Tensor, prefix: str = "flow_net.") -> Tensor:
    """SimpleFlowNet.forward, simple_flow_net.py:99-108.  x (B,D), t (B,) -> v (B,D)."""
    h = F.linear(x, sd[prefix + "input_proj.weight"], sd[prefix + "input_proj.bias"])
    te = flow_net_time_embed(sd, t, prefix)
    for l in range(flow_net_num_layers(sd, prefix)):
        h = flow_net_layer(sd, h, te, l, prefix)
    H = h.shape[-1]
    h = F.layer_norm(h, (H,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5)  # :81
    return F.linear(h, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


# ----------------------------------------------------------------------------------------
# TorchFlowMatcher  (flare/flow/flow_matchers.py)
# ----------------------------------------------------------------------------------------
def euler_sample(sd: SD, x0: Tensor, num_steps: int, return_traces: bool = False,
                 prefix: str = "flow_net."):
    """TorchFlowMatcher.sample, flow_matchers.py:55-79 with `start` given.
    dt = 1.0/K (python float); per step ts = ones(B) * (k/K); x = x + v*dt (mul, then add)."""
    x = x0
    dt = 1.0 / num_steps
    traj = [x.clone()]
    vel = []
    for k in range(num_steps):
        ts = torch.ones(x.shape[0], dtype=x.dtype) * (k / num_steps)
        v = flow_net_forward(sd, x, ts, prefix)
        x = x + v * dt
        if return_traces:
            traj.append(x.clone())
            vel.append(v.clone())
    if return_traces:
        return x, (traj, vel)
    return x


def cfm_sample_location_and_conditional_flow(x0: Tensor, x1: Tensor, t: Tensor,
                                             sigma: float = 0.0, eps: Tensor | None = None, kind: str = "conditional"):
    """torchcfm==1.0.5 sample_location_and_conditional_flow of ConditionalFlowMatcher and its two
    Gaussian-path subclasses the reference registers (RESTATED, UNPINNED; called at
    flow_matchers.py:34; classes built at :82-99):
      conditional / exact: mu_t = t*x1 + (1-t)*x0, sigma_t = sigma,                 ut = x1 - x0
      target:              mu_t = t*x1,            sigma_t = 1 - (1-sigma)*t,       ut = (x1 - (1-sigma)*xt) / (1 - (1-sigma)*t)
      schrodinger:         mu_t as conditional,    sigma_t = sigma*sqrt(t*(1-t)),
                           ut = (1-2t) / (2*t*(1-t) + 1e-8) * (xt - mu_t) + x1 - x0
    xt = mu_t + sigma_t*eps.  `t` (B,) and `eps` (= randn_like(x0)) are explicit inputs here (the
    reference draws them from the global torch RNG)."""
    tt = t.reshape(-1, *([1] * (x0.dim() - 1)))
    if kind in ("conditional", "exact"):
        mu_t = tt * x1 + (1 - tt) * x0
        xt = mu_t if (sigma == 0.0 or eps is None) else mu_t + sigma * eps
        ut = x1 - x0
    elif kind == "target":
        mu_t = tt * x1
        sigma_t = 1 - (1 - sigma) * tt
        xt = mu_t + sigma_t * eps
        ut = (x1 - (1 - sigma) * xt) / (1 - (1 - sigma) * tt)
    elif kind == "schrodinger":
        if sigma <= 0:
            raise ValueError(f"Sigma must be strictly positive, got {sigma}.")
        mu_t = tt * x1 + (1 - tt) * x0
        sigma_t = sigma * torch.sqrt(tt * (1 - tt))
        xt = mu_t + sigma_t * eps
        ratio = (1 - 2 * tt) / (2 * tt * (1 - tt) + 1e-8)
        ut = ratio * (xt - mu_t) + x1 - x0
    else:
        raise ValueError(f"unknown matcher kind {kind}")
    return t, xt, ut


def fm_loss(sd: SD, start: Tensor, target: Tensor, t: Tensor, pair_i: Tensor | None = None,
            pair_j: Tensor | None = None, sigma: float = 0.0, eps: Tensor | None = None,
            prefix: str = "flow_net.", kind: str = "conditional") -> Tensor:
    """TorchFlowMatcher.compute_loss, flow_matchers.py:19-39.  The exact-OT matcher re-pairs
    (x0[i], x1[j]) from a minibatch OT plan before the CFM draw (torchcfm OTPlanSampler,
    UNPINNED): the pairing is an explicit input (pair_i, pair_j); None = identity pairing,
    which is the `conditional` matcher."""
    x0, x1 = start, target
    if pair_i is not None:
        x0 = x0[pair_i]
    if pair_j is not None:
        x1 = x1[pair_j]
    _, xt, ut = cfm_sample_location_and_conditional_flow(x0, x1, t, sigma, eps, kind)
    vt = flow_net_forward(sd, xt, t, prefix)
    return torch.mean((vt - ut) ** 2)                      # :36


# ----------------------------------------------------------------------------------------
# encoders / decoder  (flare/networks/vita/action_ae.py, vita_policy.py:77)
# ----------------------------------------------------------------------------------------
def obs_encode(sd: SD, obs_features: Tensor) -> Tensor:
    """vita_policy.py:77,114: nn.Linear(O, D)."""
    return F.linear(obs_features, sd["obs_encoder.weight"], sd["obs_encoder.bias"])


# Test hooks for ReLU-kink analysis (tests/test_gpu_meanflow_train.py): CNN_PREACT_TRACE, when a list, receives every conv
# layer's pre-activation; CNN_RELU_MASK, when a dict {conv index: bool tensor}, replaces relu(z) by z * mask for that layer
# (identical wherever mask == (z > 0)).  Both None in normal use.
CNN_PREACT_TRACE = None
CNN_RELU_MASK = None


def cnn_action_encode(sd: SD, actions: Tensor, prefix: str = "action_encoder.") -> Tensor:
    """CNNActionEncoder.forward, action_ae.py:318-327 (ctor :283-316): transpose ->
    n x [Conv1d(k5,s2,p2)+ReLU] -> flatten -> latent_proj."""
    B = actions.shape[0]
    x = actions.transpose(1, 2)
    i = 0
    while f"{prefix}encoder.{i}.weight" in sd:
        x = F.conv1d(x, sd[f"{prefix}encoder.{i}.weight"], sd[f"{prefix}encoder.{i}.bias"],
                     stride=2, padding=2)
        if CNN_PREACT_TRACE is not None:
            CNN_PREACT_TRACE.append(x.detach())
        if CNN_RELU_MASK is not None and i in CNN_RELU_MASK:
            x = x * CNN_RELU_MASK[i].to(x.dtype)
        else:
            x = F.relu(x)
        i += 2
    x = x.reshape(B, -1)
    return F.linear(x, sd[prefix + "latent_proj.weight"], sd[prefix + "latent_proj.bias"])


def simple_action_encode(sd: SD, actions: Tensor, prefix: str = "action_encoder.") -> Tensor:
    """SimpleActionEncoder.forward, action_ae.py:161-170 (non-attention branch)."""
    B = actions.shape[0]
    x = F.linear(actions, sd[prefix + "input_proj.weight"], sd[prefix + "input_proj.bias"])
    i = 0
    while f"{prefix}layers.{i}.fc1.weight" in sd:
        x = timm_mlp(x, sd, f"{prefix}layers.{i}.", gelu_exact)
        i += 1
    return x.reshape(B, -1)


def simple_action_decode(sd: SD, z: Tensor, pred_horizon: int, action_dim: int,
                         prefix: str = "action_decoder.") -> Tensor:
    """SimpleActionDecoder.forward, action_ae.py:267-279: input_proj; n x Mlp(exact GELU, no
    residual, no norm); output_proj; view(B, T, A)."""
    x = F.linear(z, sd[prefix + "input_proj.weight"], sd[prefix + "input_proj.bias"])
    i = 0
    while f"{prefix}layers.{i}.fc1.weight" in sd:
        x = timm_mlp(x, sd, f"{prefix}layers.{i}.", gelu_exact)
        i += 1
    x = F.linear(x, sd[prefix + "output_proj.weight"], sd[prefix + "output_proj.bias"])
    return x.view(-1, pred_horizon, action_dim)


def vae_cnn_action_encode(sd: SD, actions: Tensor, eps: Tensor | None, prefix: str = "action_encoder."):
    """Variational CNNActionEncoder.forward, action_vae.py:217-229: the same conv trunk with a (2*latent_dim)-wide
    latent_proj, then DiagonalGaussianDistribution (models/dists.py:4-21): mean, logvar = chunk(params, 2, dim=1);
    sample = mean + exp(0.5*logvar) * eps, eps = randn_like(mean) (an input here); eps None = deterministic (:12-18):
    sample = mean.  Returns (mean, logvar, z)."""
    params = cnn_action_encode(sd, actions, prefix)
    mean, logvar = torch.chunk(params, 2, dim=1)
    if eps is None:
        return mean, logvar, mean
    std = torch.exp(0.5 * logvar)
    return mean, logvar, mean + std * eps


def is_variational(sd: SD) -> bool:
    w = sd.get("action_encoder.latent_proj.weight")
    return w is not None and w.shape[0] == 2 * sd["obs_encoder.weight"].shape[0]


def action_encode(sd: SD, actions: Tensor) -> Tensor:
    if "action_encoder.latent_proj.weight" in sd:
        return cnn_action_encode(sd, actions)
    return simple_action_encode(sd, actions)


# ----------------------------------------------------------------------------------------
# normalisation  (flare/utils/normalize.py)
# ----------------------------------------------------------------------------------------
def normalize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Normalize.forward, normalize.py:38-46."""
    return (x - mean) / (std + 1e-8)


def unnormalize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Unnormalize.forward, normalize.py:62-70."""
    return x * std + mean


# ----------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------
def contrastive_loss(image_features: Tensor, action_features: Tensor, temperature: float = 0.07):
    """compute_contrastive_loss, contrastive_loss.py:5-22 (symmetric InfoNCE)."""
    B = image_features.size(0)
    a = F.normalize(image_features, dim=1)
    b = F.normalize(action_features, dim=1)
    logits = torch.matmul(a, b.T) / temperature
    labels = torch.arange(B)
    return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.T, labels)) / 2


def gaussian_kl(mean: Tensor, logvar: Tensor) -> Tensor:
    """DiagonalGaussianDistribution.kl(other=None), dists.py:23-31, then `.mean()` as at
    vita_policy.py:148."""
    dims = list(range(1, mean.dim()))
    kl = 0.5 * torch.sum(torch.pow(mean, 2) + torch.exp(logvar) - 1.0 - logvar, dim=dims)
    return kl.mean()


# ----------------------------------------------------------------------------------------
# alternate matchers, inference side  
# ----------------------------------------------------------------------------------------
def consistency_sample(sd: SD, x0: Tensor, num_steps: int, eps: float = 1e-2, sigma_var: float = 1.0,
                       noise_scale: float = 1.0, noise: Tensor | None = None, prefix: str = "flow_net.") -> Tensor:
    """ConsistencyFlowMatcher.sample, consistency_flow_matcher.py:72-106 with `start` given.  `noise[i]`
    stands for the torch.randn_like draw of step i (:96)."""
    z = x0.detach().clone()
    dt = 1.0 / num_steps
    for i in range(num_steps):
        num_t = i / num_steps * (1 - eps) + eps
        t = torch.ones(z.shape[0], dtype=z.dtype) * num_t
        vt = flow_net_forward(sd, z, t, prefix)
        sigma_t = (1.0 - num_t) * sigma_var
        if sigma_t > 0:
            pred_sigma = vt + (sigma_t ** 2) / (2 * (noise_scale ** 2) * ((1 - num_t) ** 2)) * \
                (0.5 * num_t * (1 - num_t) * vt - 0.5 * (2 - num_t) * z)
            z = z + pred_sigma * dt + sigma_t * math.sqrt(dt) * noise[i]
        else:
            z = z + vt * dt
    return z


def mean_flow_num_layers(sd: SD, prefix: str = "flow_net.") -> int:
    n = 0
    while f"{prefix}hidden_layers.{n}.0.weight" in sd:
        n += 1
    return n


def mean_flow_net_forward(sd: SD, x: Tensor, timestep: Tensor, h: Tensor, prefix: str = "flow_net."):
    """SimpleMeanFlowNet.forward, simple_mean_flow_net.py:63-81 (dropout 0) -> (u, v, internal_features)."""
    def lin(name, y):
        return F.linear(y, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"])
    t_emb = lin("t_embed.2", F.silu(lin("t_embed.0", timestep.unsqueeze(-1))))
    h_emb = lin("h_embed.2", F.silu(lin("h_embed.0", h.unsqueeze(-1))))
    time_cond = t_emb + h_emb
    x = lin("input_layer", x)
    feats = []
    for l in range(mean_flow_num_layers(sd, prefix)):
        x_cond = x + time_cond
        x = lin(f"hidden_layers.{l}.3", F.silu(lin(f"hidden_layers.{l}.0", x_cond))) + x
        feats.append(x)
    return lin("u_output_layer", x), lin("v_output_layer", x), feats


def mean_flow_sample(sd: SD, start: Tensor, num_steps: int = 1, prefix: str = "flow_net.") -> Tensor:
    """MeanFlowMatcher.sample, mean_flow_matcher.py:201-233."""
    x = start
    B = x.shape[0]
    for step in range(num_steps):
        t_scalar = 1.0 - step / num_steps
        r_scalar = 1.0 - (step + 1) / num_steps
        t = torch.full((B,), t_scalar, dtype=x.dtype)
        h = torch.full((B,), t_scalar - r_scalar, dtype=x.dtype)
        u, _, _ = mean_flow_net_forward(sd, x, t, h, prefix)
        x = x - h.view(-1, 1) * u
    return x


def dispersive_loss(z: Tensor, tau: float = 1.0):
    """dispersive_loss, mean_flow_matcher.py:27-39."""
    if z.shape[0] <= 1:
        return 0.0
    dist = torch.cdist(z, z, p=2) ** 2
    dist = dist / (torch.max(dist).detach() + 1e-8)
    return torch.log(torch.mean(torch.exp(-dist / tau)))


def mean_flow_dispersive_total(sd: SD, start: Tensor, target: Tensor, t: Tensor, r: Tensor, tau: float = 1.0,
                               prefix: str = "flow_net."):
    """Sum of dispersive_loss over the hidden layers' outputs at (z_t, t, r), mean_flow_matcher.py:185-192."""
    t_ = t.view(-1, *([1] * (start.dim() - 1)))
    z_t = (1 - t_) * target + t_ * start
    feats = mean_flow_net_forward(sd, z_t, t, t - r, prefix)[2]
    total = 0.0
    for f in feats:
        total = total + dispersive_loss(f, tau)
    return total


def improved_mean_flow_loss(sd: SD, start: Tensor, target: Tensor, t: Tensor, r: Tensor, norm_p: float = 1.0, norm_eps: float = 0.01,
                            aux_v_loss_weight: float = 1.0, prefix: str = "flow_net."):
    """MeanFlowMatcher.compute_loss with use_imf = True, mean_flow_matcher.py:137-163 (adaptive_imf_loss :21-24).
    Returns (loss, metrics)."""
    x1, x0 = target, start
    t_ = t.view(-1, *([1] * (x0.dim() - 1)))
    h_ = (t - r).view(-1, *([1] * (x0.dim() - 1)))
    z_t = (1 - t_) * x1 + t_ * x0
    v = x0 - x1
    with torch.no_grad():                                                         # :138-140
        v_net = mean_flow_net_forward(sd, z_t, t, t - t, prefix)[1]

    def pred_imf(z_in, t_in, r_in):
        u_, v_, _ = mean_flow_net_forward(sd, z_in, t_in, t_in - r_in, prefix)
        return u_, v_

    (u, v_pred), (dudt, _) = torch.autograd.functional.jvp(pred_imf, (z_t, t, r), (v_net, torch.ones_like(t), torch.zeros_like(r)),
                                                          create_graph=True)     # :146-151
    compound = u + h_ * dudt.detach()

    def adaptive_imf(error):
        per_sample = torch.sum(error ** 2, dim=tuple(range(1, error.ndim)))
        return (per_sample / (per_sample + norm_eps).pow(norm_p).detach()).mean()

    imf_loss = adaptive_imf(compound - v.detach())
    aux_v_loss = adaptive_imf(v_pred - v.detach())
    loss = imf_loss + aux_v_loss_weight * aux_v_loss
    return loss, {"imf_loss": imf_loss, "aux_v_loss": aux_v_loss, "imf_mse": torch.mean((compound - v) ** 2),
                  "aux_v_mse": torch.mean((v_pred - v) ** 2)}


def mean_flow_loss(sd: SD, start: Tensor, target: Tensor, t: Tensor, r: Tensor, adaptive_loss_gamma: float = 0.5,
                   c: float = 1e-3, prefix: str = "flow_net.") -> Tensor:
    """MeanFlowMatcher.compute_loss, mean_flow_matcher.py:109-183 with use_imf = False and dispersive_loss_weight = 0;
    (t, r) — the reference's sample_t_r draw (:87-107) — are explicit inputs.  adaptive_l2_loss: :10-18."""
    x1, x0 = target, start
    t_ = t.view(-1, *([1] * (x0.dim() - 1)))
    h_ = (t - r).view(-1, *([1] * (x0.dim() - 1)))
    z_t = (1 - t_) * x1 + t_ * x0                                                 # :129
    v = x0 - x1                                                                   # :132

    def pred_u(z_in, t_in, r_in):
        return mean_flow_net_forward(sd, z_in, t_in, t_in - r_in, prefix)[0]

    u, dudt = torch.autograd.functional.jvp(pred_u, (z_t, t, r), (v, torch.ones_like(t), torch.zeros_like(r)), create_graph=True)  # :171-176
    u_tgt = v - h_ * dudt                                                         # :178
    error = u - u_tgt.detach()
    delta_sq = torch.mean(error ** 2, dim=tuple(range(1, error.ndim)))
    w = 1.0 / (delta_sq + c).pow(1.0 - adaptive_loss_gamma)
    return (w.detach() * delta_sq).mean()


def mean_flow_sample_grad(sd: SD, start: Tensor, num_steps: int = 1, prefix: str = "flow_net.") -> Tensor:
    """MeanFlowMatcher.sample (:201-233) keeping the graph (vita_policy.py:154-160 calls it with grad)."""
    x = start
    B = x.shape[0]
    for step in range(num_steps):
        t_scalar = 1.0 - step / num_steps
        r_scalar = 1.0 - (step + 1) / num_steps
        t = torch.full((B,), t_scalar, dtype=x.dtype)
        h = torch.full((B,), t_scalar - r_scalar, dtype=x.dtype)
        u = mean_flow_net_forward(sd, x, t, h, prefix)[0]
        x = x - h.view(-1, *([1] * (x.dim() - 1))) * u
    return x


def compute_loss_meanflow(sd: SD, obs_features: Tensor, gt_actions: Tensor, t: Tensor, r: Tensor, num_steps: int = 1,
                          consistency_weight: float = 1.0, enc_recon_weight: float = 0.5, flow_recon_weight: float = 0.5,
                          recon_loss_type: str = "l1", decode_flow_latents: bool = True, adaptive_loss_gamma: float = 0.5,
                          use_imf: bool = False, norm_p: float = 1.0, norm_eps: float = 0.01, aux_v_loss_weight: float = 1.0,
                          dispersive_loss_weight: float = 0.0, dispersive_loss_tau: float = 1.0,
                          freeze_encoder: bool = False, freeze_decoder: bool = False,
                          enc_contrastive_weight: float = 0.0, flow_contrastive_weight: float = 0.0,
                          kl_weight: float = 0.0, enc_noise: Tensor | None = None):
    """VitaPolicy.compute_loss (vita_policy.py:112-189) with flow_matcher.name = 'mean' over SimpleMeanFlowNet: defaults
    otherwise (plain autoencoder, nothing frozen, no contrastive terms).  Returns (loss, metrics of 0-dim tensors)."""
    recon = F.l1_loss if recon_loss_type == "l1" else F.mse_loss
    T, A = gt_actions.shape[1], gt_actions.shape[2]
    m = {}
    obs_latents = obs_encode(sd, obs_features)
    vae = is_variational(sd)
    if vae:                                                                                        # :121
        post_mean, post_logvar, action_latents = vae_cnn_action_encode(sd, gt_actions, enc_noise)
    else:
        action_latents = action_encode(sd, gt_actions)
    if freeze_encoder:                                                                             # :119
        action_latents = action_latents.detach()
    if use_imf:
        flow_loss, fm = improved_mean_flow_loss(sd, obs_latents, action_latents, t, r, norm_p, norm_eps, aux_v_loss_weight)
        m.update(fm)
    else:
        flow_loss = mean_flow_loss(sd, obs_latents, action_latents, t, r, adaptive_loss_gamma)    # :126-130
        m["meanflow_loss"] = flow_loss
    if dispersive_loss_weight > 0:                                                                 # mean_flow_matcher.py:188-195
        dis = mean_flow_dispersive_total(sd, obs_latents, action_latents, t, r, dispersive_loss_tau)
        m["dispersive_loss"] = dis
        flow_loss = flow_loss + dispersive_loss_weight * dis
    loss = flow_loss
    m["flow_loss"] = flow_loss
    Bn = obs_latents.shape[0]
    if enc_contrastive_weight > 0:                                                                 # :134-139
        c = contrastive_loss(obs_latents.view(Bn, -1), action_latents.view(Bn, -1))
        loss = loss + enc_contrastive_weight * c
        m["enc_contrastive_loss"] = c
    if freeze_encoder and freeze_decoder:                                                          # :142-143
        m["loss"] = loss
        return loss, m
    if not freeze_encoder and vae and kl_weight > 0:                                               # :146-150
        kl = torch.zeros((), dtype=obs_latents.dtype) if enc_noise is None else gaussian_kl(post_mean, post_logvar)
        m["action_kl_loss"] = kl
        loss = loss + kl_weight * kl
    if decode_flow_latents and not freeze_encoder and not freeze_decoder:                          # :153-178
        pred = mean_flow_sample_grad(sd, obs_latents, num_steps)
        if consistency_weight > 0:
            cterm = F.mse_loss(pred, action_latents)
            loss = loss + consistency_weight * cterm
            m["consistency_loss"] = cterm
        if flow_contrastive_weight > 0:                                                            # :167-172
            c = contrastive_loss(obs_latents.view(Bn, -1), pred.view(Bn, -1))
            loss = loss + flow_contrastive_weight * c
            m["flow_contrastive_loss"] = c
        if flow_recon_weight > 0:
            rterm = recon(simple_action_decode(sd, pred, T, A), gt_actions)
            m["flow_action_recon_loss"] = rterm
            loss = loss + flow_recon_weight * rterm
    if enc_recon_weight > 0 and not freeze_decoder:                                                # :183-187
        rterm = recon(simple_action_decode(sd, action_latents, T, A), gt_actions)
        m["enc_action_recon_loss"] = rterm
        loss = loss + enc_recon_weight * rterm
    m["loss"] = loss
    return loss, m


def make_meanflow_state_dict(latent_dim: int = 512, hidden_dim: int = 512, num_layers: int = 4, mlp_ratio: float = 4.0,
                             seed: int = 0, dtype=torch.float32, prefix: str = "flow_net.") -> SD:
    """Seeded weights with SimpleMeanFlowNet's keys; the output layers are NOT zero (the reference zero-initialises
    them, which would make every parity check trivial)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def lin(name, out_f, in_f, bias_std=0.05):
        bound = math.sqrt(6.0 / (in_f + out_f))
        sd[prefix + name + ".weight"] = ((torch.rand(out_f, in_f, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
        sd[prefix + name + ".bias"] = (torch.randn(out_f, generator=g, dtype=torch.float64) * bias_std).to(dtype)
    ff = int(hidden_dim * mlp_ratio)
    lin("t_embed.0", hidden_dim, 1); lin("t_embed.2", hidden_dim, hidden_dim)
    lin("h_embed.0", hidden_dim, 1); lin("h_embed.2", hidden_dim, hidden_dim)
    lin("input_layer", hidden_dim, latent_dim)
    for l in range(num_layers):
        lin(f"hidden_layers.{l}.0", ff, hidden_dim)
        lin(f"hidden_layers.{l}.3", hidden_dim, ff)
    lin("u_output_layer", latent_dim, hidden_dim)
    lin("v_output_layer", latent_dim, hidden_dim)
    return sd


# ----------------------------------------------------------------------------------------
# policy-level entry points  (flare/policies/vita/vita_policy.py)
# ----------------------------------------------------------------------------------------
def generate_actions(sd: SD, obs_features: Tensor, num_steps: int, pred_horizon: int,
                     action_dim: int) -> Tensor:
    """VitaPolicy.generate_actions, vita_policy.py:191-208, from observer features on:
    obs_encoder -> FM.sample(start=obs_latents) -> action_decoder."""
    z0 = obs_encode(sd, obs_features)
    z = euler_sample(sd, z0, num_steps)
    return simple_action_decode(sd, z, pred_horizon, action_dim)


def cnn_action_decode(sd: SD, z: Tensor, pred_horizon: int, hidden_dim: int, prefix: str = "action_decoder.") -> Tensor:
    """CNNActionDecoder.forward, action_vae.py:284-298: latent_proj, view as (B, hidden, L0), ConvTranspose1d(k5, s2, p2,
    output_padding 1) with ReLU between (none after the last), transpose to (B, T, A); linear interpolation to pred_horizon when
    the length differs (:293-296)."""
    x = F.linear(z, sd[prefix + "latent_proj.weight"], sd[prefix + "latent_proj.bias"])
    x = x.view(z.shape[0], hidden_dim, -1)
    n = 0
    while f"{prefix}decoder.{2 * n}.weight" in sd:
        n += 1
    for i in range(n):
        x = F.conv_transpose1d(x, sd[f"{prefix}decoder.{2 * i}.weight"], sd[f"{prefix}decoder.{2 * i}.bias"], stride=2, padding=2,
                               output_padding=1)
        if i < n - 1:
            x = F.relu(x)
    if x.shape[2] != pred_horizon:
        x = F.interpolate(x, size=pred_horizon, mode="linear", align_corners=False)
    return x.transpose(1, 2)


def compute_loss(sd: SD, obs_features: Tensor, gt_actions: Tensor, t: Tensor, num_steps: int,
                 pair_i: Tensor | None = None, pair_j: Tensor | None = None,
                 consistency_weight: float = 1.0, enc_recon_weight: float = 0.5,
                 flow_recon_weight: float = 0.5, recon_loss_type: str = "l1",
                 enc_contrastive_weight: float = 0.0, flow_contrastive_weight: float = 0.0,
                 decode_flow_latents: bool = True, sigma: float = 0.0,
                 freeze_encoder: bool = False, freeze_decoder: bool = False,
                 kl_weight: float = 0.0, enc_noise: Tensor | None = None,
                 fm_kind: str = "conditional", fm_noise: Tensor | None = None, enc_latents: Tensor | None = None,
                 decoder=None):
    """VitaPolicy.compute_loss, vita_policy.py:112-189, from observer features on, with the
    (enc_latents: the action latents of an encoder that is not in `sd` — another encoder_type's module — instead of
    action_encode(sd, ...); everything downstream is unchanged)
    (decoder: a callable latents (B, D) -> actions (B, T, A) that is not in `sd` — another decoder_type's module, e.g.
    cnn_action_decode below — instead of simple_action_decode(sd, ...))
    freeze flags of :119,142-143,153,174,183.  A state dict whose action_encoder.latent_proj is
    2*latent_dim wide is the variational autoencoder (use_variational, :120-121,146-150):
    `enc_noise` is the posterior's randn_like draw (None = eval mode's deterministic posterior).
    Returns (loss, metrics) where metrics holds 0-dim tensors (the reference calls .item()
    on each)."""
    recon = F.l1_loss if recon_loss_type == "l1" else F.mse_loss
    T, A = gt_actions.shape[1], gt_actions.shape[2]
    B = obs_features.shape[0]
    m = {}
    decode = decoder if decoder is not None else (lambda z: simple_action_decode(sd, z, T, A))
    obs_latents = obs_encode(sd, obs_features)                                  # :114
    vae = enc_latents is None and is_variational(sd)
    if enc_latents is not None:
        action_latents = enc_latents
    elif vae:                                                                   # :121
        post_mean, post_logvar, action_latents = vae_cnn_action_encode(sd, gt_actions, enc_noise)
    else:
        action_latents = action_encode(sd, gt_actions)                          # :123
    if freeze_encoder:                                                          # :119 torch.no_grad()
        action_latents = action_latents.detach()
    flow_loss = fm_loss(sd, obs_latents, action_latents, t, pair_i, pair_j, sigma, fm_noise, kind=fm_kind)  # :126-130
    loss = flow_loss
    m["flow_loss"] = flow_loss
    if enc_contrastive_weight > 0:                                              # :134-139
        c = contrastive_loss(obs_latents.view(B, -1), action_latents.view(B, -1))
        loss = loss + enc_contrastive_weight * c
        m["enc_contrastive_loss"] = c
    if freeze_encoder and freeze_decoder:                                       # :142-143
        m["loss"] = loss
        return loss, m
    if not freeze_encoder and vae and kl_weight > 0:                            # :146-150
        if enc_noise is None:                                                   # dists.py:25-26
            kl = torch.zeros((), dtype=obs_latents.dtype)
        else:
            kl = gaussian_kl(post_mean, post_logvar)
        m["action_kl_loss"] = kl
        loss = loss + kl_weight * kl
    if decode_flow_latents and not freeze_encoder and not freeze_decoder:       # :153-178
        pred = euler_sample(sd, obs_latents, num_steps)
        if consistency_weight > 0:
            c = F.mse_loss(pred, action_latents)
            loss = loss + consistency_weight * c
            m["consistency_loss"] = c
        if flow_contrastive_weight > 0:
            c = contrastive_loss(obs_latents.view(B, -1), pred.view(B, -1))
            loss = loss + flow_contrastive_weight * c
            m["flow_contrastive_loss"] = c
        if flow_recon_weight > 0 and not freeze_decoder:                        # :174
            r = recon(decode(pred), gt_actions)
            m["flow_action_recon_loss"] = r
            loss = loss + flow_recon_weight * r
    if enc_recon_weight > 0 and not freeze_decoder:                             # :183-187
        r = recon(decode(action_latents), gt_actions)
        m["enc_action_recon_loss"] = r
        loss = loss + enc_recon_weight * r
    m["loss"] = loss
    return loss, m


# ----------------------------------------------------------------------------------------
# trainer tail  (flare/runners/policy_trainer.py:66-102; vita_policy.py:222-237)
# ----------------------------------------------------------------------------------------
def cosine_with_warmup_lambda(step: int, num_warmup_steps: int, num_training_steps: int,
                              num_cycles: float = 0.5) -> float:
    """diffusers==0.30.3 get_cosine_schedule_with_warmup lr_lambda (RESTATED, UNPINNED)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def ema_decay(optimization_step: int, max_decay: float = 0.9999, min_decay: float = 0.0,
              inv_gamma: float = 1.0, power: float = 0.75, use_ema_warmup: bool = False) -> float:
    """diffusers==0.30.3 EMAModel.get_decay (RESTATED, UNPINNED).  The reference passes only
    `power` (base_policy.py:45-49) so use_ema_warmup is False and decay = (1+s)/(10+s)."""
    step = max(0, optimization_step - 0 - 1)
    if step <= 0:
        return 0.0
    if use_ema_warmup:
        cur = 1 - (1 + step / inv_gamma) ** -power
    else:
        cur = (1 + step) / (10 + step)
    return max(min(cur, max_decay), min_decay)


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1: float = 0.95, beta2: float = 0.999, eps: float = 1e-8,
               weight_decay: float = 1e-6):
    """torch.optim.AdamW single-tensor update (the optimizer of vita_policy.py:222-229);
    `step` is 1-based.  Updates p, m, v in place."""
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# ----------------------------------------------------------------------------------------
# deterministic synthetic weights: vita_amd/synthetic.py (re-exported here for the tests)
# ----------------------------------------------------------------------------------------
from vita_amd.synthetic import make_state_dict  # noqa: E402,F401


def cast_sd(sd: SD, dtype) -> SD:
    return {k: v.to(dtype) for k, v in sd.items()}
