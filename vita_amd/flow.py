"""Host-side mirror of flare/flow/flow_matchers.py for the torchcfm-backed matchers.

Same names, constructor arguments and error behaviour as the reference (`get_flow_matcher(name=..., sigma=...,
num_sampling_steps=...)`, `FLOW_MATCHER_CLASSES`, `.sample(model, shape, device, num_steps, return_traces,
start)`, `.compute_loss(model, target, start)`), but `model` is a `vita_amd.ops.FlowNetHandle` and the arithmetic
runs in libvita_hip.so: `.sample` is ONE C-ABI call that enqueues all K Euler steps.

All four torchcfm-backed names are built (conditional, exact, target, schrodinger; any sigma >= 0).
What stays on the host, as in the reference: the flow-matching draw.  torchcfm==1.0.5 (absent from this image;
restated from its published source, UNPINNED) draws t with `torch.rand(B).type_as(x0)` on the CPU generator and,
for the `exact` matcher, solves a minibatch OT plan with POT's `ot.emd` on numpy and samples B pairs from it
with `np.random.choice(..., replace=True)`.  Here the plan comes from scipy's `linear_sum_assignment` (for uniform
marginals of equal size the optimal plan is a permutation / B, which is what the network simplex returns up to
ties), sampled with the same numpy call sequence.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import ops


def draw_t(batch_size: int, like: torch.Tensor) -> torch.Tensor:
    """torchcfm ConditionalFlowMatcher.sample_location_and_conditional_flow: t = torch.rand(B).type_as(x0)."""
    return torch.rand(batch_size).to(device=like.device, dtype=like.dtype)


def exact_ot_pairing(x0: torch.Tensor, x1: torch.Tensor):
    """torchcfm OTPlanSampler(method='exact').sample_plan: M = cdist(x0, x1)**2 -> ot.emd(a, b, M) -> sample_map.
    Returns (i, j) int64 index arrays on the CPU, drawn with replacement (an index may repeat)."""
    from scipy.optimize import linear_sum_assignment
    B = x0.shape[0]
    a = x0.reshape(B, -1).double()
    b = x1.reshape(B, -1).double()
    # squared euclidean cost; torch.cdist here is host-side LP set-up exactly as in torchcfm, not the kernel path
    M = (torch.cdist(a, b) ** 2).cpu().numpy()
    rows, cols = linear_sum_assignment(M)
    pi = np.zeros((B, B))
    pi[rows, cols] = 1.0 / B
    p = pi.flatten()
    p = p / p.sum()
    choices = np.random.choice(pi.shape[0] * pi.shape[1], p=p, size=B, replace=True)
    i, j = np.divmod(choices, pi.shape[1])
    return torch.from_numpy(i.astype(np.int64)), torch.from_numpy(j.astype(np.int64))


class TorchFlowMatcher:
    """flow_matchers.py:10-79 over torchcfm==1.0.5's ConditionalFlowMatcher family (restated, UNPINNED)."""

    pairing = "identity"
    kind = L.FM_CONDITIONAL
    default_sigma = 0.0

    def __init__(self, num_sampling_steps: int = 6, sigma=None, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected flow matcher arguments: {sorted(kwargs)}")
        self.sigma = float(self.default_sigma if sigma is None else sigma)
        if self.sigma < 0.0:
            raise ValueError(f"sigma must be >= 0, got {self.sigma}")
        self.num_sampling_steps = num_sampling_steps

    @property
    def needs_noise(self) -> bool:
        """Whether xt depends on the eps = randn_like(x0) draw (torchcfm always draws it; with the conditional path and
        sigma = 0 it is multiplied by zero, so the draw is skipped here)."""
        return self.kind != L.FM_CONDITIONAL or self.sigma != 0.0

    # -- the draw (host) ---------------------------------------------------------------
    def draw(self, x0: torch.Tensor, x1: torch.Tensor):
        """(t, pair_i, pair_j, eps) for one compute_loss call in torchcfm's order: OT plan (numpy RNG), t = torch.rand (CPU
        generator), eps = randn_like(x0) (device generator).  pair_* are None for the identity pairing, eps None when unused."""
        pi = pj = None
        if self.pairing == "exact":
            pi, pj = exact_ot_pairing(x0.detach(), x1.detach())
        t = draw_t(x0.shape[0], x0)
        eps = torch.randn_like(x0) if self.needs_noise else None
        return t, pi, pj, eps

    # -- reference API -----------------------------------------------------------------
    def compute_loss(self, model: "ops.FlowNetHandle", target: torch.Tensor, start: Optional[torch.Tensor] = None, **kwargs):
        """flow_matchers.py:19-39 (forward only; training uses the fused vita_compute_loss_f32)."""
        x0 = torch.randn_like(target) if start is None else start
        t, pi, pj, eps = self.draw(x0, target)
        xt, ut = ops.fm_interpolate(self.kind, self.sigma, x0, target, t, eps, pair_j=pj, pair_i=pi)
        vt = ops.flownet_forward(model, xt, t)
        loss = ops.mse_loss_fwd(vt, ut)
        return loss, {"loss": loss.item()}

    def sample(self, model: "ops.FlowNetHandle", shape, device, num_steps=None, return_traces=False, start=None, **kwargs):
        """flow_matchers.py:41-79: one C-ABI call for all Euler steps."""
        if num_steps is None:
            num_steps = self.num_sampling_steps
        x = torch.randn(shape, device=device) if start is None else start
        if not return_traces:
            return ops.ode_sample(model, x, num_steps)
        out, traj = ops.ode_sample(model, x, num_steps, return_traces=True)
        # the reference returns per-step CPU copies of x and v (:63-66,73-78); v_k = (x_{k+1} - x_k) / dt is not
        # recomputed here: callers (visualize) only use the latent history
        traj_history = [x] + [traj[k + 1].cpu() for k in range(num_steps)]
        vel_history = [np.zeros_like(x.cpu())] + [((traj[k + 1] - traj[k]) * float(num_steps)).cpu() for k in range(num_steps)]
        return out, (traj_history, vel_history)


class ConditionalFlowMatcher(TorchFlowMatcher):
    pairing = "identity"


class ExactOptimalTransportConditionalFlowMatcher(TorchFlowMatcher):
    pairing = "exact"


class TargetConditionalFlowMatcher(TorchFlowMatcher):
    """flow_matchers.py:87-89 -> torchcfm TargetConditionalFlowMatcher (Lipman et al.): mu_t = t*x1,
    sigma_t = 1 - (1-sigma)*t, ut = (x1 - (1-sigma)*xt) / (1 - (1-sigma)*t); x0 only lends its shape."""
    kind = L.FM_TARGET


class SchrodingerBridgeConditionalFlowMatcher(TorchFlowMatcher):
    """flow_matchers.py:92-94 -> torchcfm SchrodingerBridgeConditionalFlowMatcher(sigma=1.0, ot_method='exact'): the
    minibatch-OT pairing of the exact matcher (ot.emd ignores the entropic `reg`), a Brownian-bridge xt and its
    conditional flow.  sigma must be > 0 (vita.yaml's flow_matcher.sigma: 0.0 has to be overridden, as in the reference)."""
    kind = L.FM_SCHRODINGER
    pairing = "exact"
    default_sigma = 1.0

    def __init__(self, num_sampling_steps: int = 6, sigma=None, ot_method: str = "exact", **kwargs):
        super().__init__(num_sampling_steps, sigma, **kwargs)
        if self.sigma <= 0:
            raise ValueError(f"Sigma must be strictly positive, got {self.sigma}.")
        if ot_method != "exact":
            raise NotImplementedError("schrodinger matcher: only ot_method='exact' (torchcfm's default) is built")


class _InferenceOnlyMatcher:
    """Matchers whose sampler is built and whose training loss is not."""

    inference_only = True
    pairing = "identity"

    needs_noise = False
    kind, sigma = L.FM_CONDITIONAL, 0.0

    def draw(self, x0, x1):
        raise NotImplementedError(f"{type(self).__name__}: the training loss of this matcher is not built")


class MeanFlowMatcher(_InferenceOnlyMatcher):
    """flow/mean_flow_matcher.py:42-233 — constructor arguments kept; `.sample` is ONE C-ABI call over a
    `vita_amd.ops.MeanFlowNetHandle` (SimpleMeanFlowNet).  Training: the MeanFlow loss (JVP target + adaptive L2 loss, :164-183) and the
    improved-MeanFlow loss (use_imf: no-grad v-head tangent, compound velocity, auxiliary v loss, :137-163) run inside the
    policy's fused MeanFlow step (vita_compute_loss_meanflow_f32) with the (t, r) draw of `sample_t_r` made here on the host as
    in the reference, and so does the dispersive term on the hidden layers' outputs (dispersive_loss_weight > 0, :27-39, 185-195)."""

    inference_only = False

    def sample_t_r(self, batch_size, device):
        """mean_flow_matcher.py:87-107, statement for statement (torch's generators, as in the reference)."""
        normal_samples = torch.randn(batch_size, 2, device=device) * self.time_dist_sigma + self.time_dist_mu
        samples = torch.sigmoid(normal_samples)
        t = torch.max(samples, dim=1)[0]
        r = torch.min(samples, dim=1)[0]
        num_selected = int(self.flow_ratio * batch_size)
        indices = torch.randperm(batch_size, device=device)[:num_selected]
        r[indices] = t[indices]
        return t, r

    def __init__(self, flow_ratio=0.5, time_dist_mu=-0.4, time_dist_sigma=1.0, adaptive_loss_gamma=0.5, norm_p=1.0, norm_eps=0.01,
                 aux_v_loss_weight=1.0, dispersive_loss_tau=1.0, dispersive_loss_weight=0.0, cfg_scale=0.5, use_imf=False, **kwargs):
        self.flow_ratio, self.time_dist_mu, self.time_dist_sigma = flow_ratio, time_dist_mu, time_dist_sigma
        self.adaptive_loss_gamma, self.norm_p, self.norm_eps = adaptive_loss_gamma, norm_p, norm_eps
        self.aux_v_loss_weight, self.dispersive_loss_tau = aux_v_loss_weight, dispersive_loss_tau
        self.dispersive_loss_weight, self.cfg_scale, self.use_imf = dispersive_loss_weight, cfg_scale, use_imf
        self.num_sampling_steps = 1

    def compute_loss(self, model, target, start=None, **kwargs):
        raise NotImplementedError("MeanFlowMatcher.compute_loss on its own is not exposed: the MeanFlow loss runs inside "
                                  "VitaPolicy.compute_loss (vita_compute_loss_meanflow_f32)")

    def sample(self, model: "ops.MeanFlowNetHandle", shape, device, num_steps=None, return_traces=False, start=None, **kwargs):
        """mean_flow_matcher.py:201-233."""
        if start is None:
            raise ValueError("MeanFlowMatcher requires a 'start' (vision latent) tensor for sampling.")      # :205-206
        if not isinstance(model, ops.MeanFlowNetHandle):
            raise TypeError("MeanFlowMatcher.sample needs a SimpleMeanFlowNet (policy.flow_net.name: simple_mean_flow_net)")
        num_steps = self.num_sampling_steps if num_steps is None else num_steps
        x = ops.meanflow_sample(model, start, num_steps)
        if not return_traces:
            return x
        # :217-230 — the per-step CPU copies of x and of the mean velocity.  The returned x is the fused sampler's (the
        # same bits as without traces); the histories re-run the steps one vita_meanflow_forward_f32 call at a time.
        xk = start.detach().float()
        traj_history, vel_history = [xk.clone().cpu()], [torch.zeros_like(xk).cpu()]
        B = xk.shape[0]
        for step in range(num_steps):
            t_scalar = 1.0 - step / num_steps
            h_scalar = t_scalar - (1.0 - (step + 1) / num_steps)
            t = torch.full((B,), t_scalar, device=xk.device)
            h = torch.full((B,), h_scalar, device=xk.device)
            u, _ = ops.meanflow_forward(model, xk, t, h, need_v=False)
            xk = xk - h.view(-1, 1) * u
            traj_history.append(xk.clone().cpu())
            vel_history.append(u.clone().cpu())
        return x, (traj_history, vel_history)


class MeanFlowConditionalFlowMatcher(MeanFlowMatcher):
    def __init__(self, num_sampling_steps=1, **kwargs):                     # flow_matchers.py:107-111
        if num_sampling_steps != 1:
            print("Warning: MeanFlow is designed for 1-NFE generation.")
        super().__init__(**kwargs)
        self.num_sampling_steps = num_sampling_steps


class ImprovedMeanFlowConditionalFlowMatcher(MeanFlowMatcher):
    def __init__(self, num_sampling_steps=1, **kwargs):                     # flow_matchers.py:119-125
        kwargs["use_imf"] = True
        if num_sampling_steps != 1:
            print("Warning: Improved MeanFlow is designed for 1-NFE generation.")
        super().__init__(**kwargs)
        self.num_sampling_steps = num_sampling_steps


class ConsistencyFlowMatcher(_InferenceOnlyMatcher):
    """flow/consistency_flow_matcher.py:7-127 + flow_matchers.py:128-132.  `.sample` is ONE C-ABI call over the
    SimpleFlowNet handle; with sigma_var > 0 (the default 1.0) the per-step standard-normal draws the reference takes from
    torch.randn_like are drawn here in one torch.randn call (or passed as `noise=`).  `.compute_loss` cannot run on VITA's
    2-D latents in the reference either (it indexes target.shape[2], :44) and is not built."""

    def __init__(self, num_sampling_steps=1, eps=1e-2, num_segments=2, boundary=1, delta=1e-3, alpha=1e-5, noise_scale=1.0,
                 sigma_var=1.0, ode_tol=1e-5):
        if num_sampling_steps != 1:
            print("Warning: ConsistencyFlow is designed for 1-NFE generation.")
        self.eps, self.num_segments, self.boundary, self.delta, self.alpha = eps, num_segments, boundary, delta, alpha
        self.noise_scale, self.sigma_var, self.ode_tol = noise_scale, sigma_var, ode_tol
        self.sigma_t = lambda t: (1. - t) * sigma_var
        self.num_sampling_steps = num_sampling_steps

    def compute_loss(self, model, target, start=None, **kwargs):
        raise NotImplementedError("Consistency-FM training is not built (and the reference's loss indexes target.shape[2], "
                                  "consistency_flow_matcher.py:44, so it does not accept VITA's (B, D) latents)")

    def sample(self, model: "ops.FlowNetHandle", shape, device, num_steps=None, return_traces=False, start=None, noise=None, **kwargs):
        if num_steps is None:
            num_steps = self.num_sampling_steps
        z = torch.randn(shape, device=device) if start is None else start
        if self.sigma_var > 0 and noise is None:
            noise = torch.randn((num_steps,) + tuple(z.shape), device=z.device, dtype=torch.float32)
        out = ops.consistency_sample(model, z, num_steps, eps=self.eps, sigma_var=self.sigma_var, noise_scale=self.noise_scale,
                                     noise=noise)
        if not return_traces:
            return out
        # :84-104 — per-step CPU copies of z and vt (no initial entry, unlike the Euler and MeanFlow samplers).  The returned
        # z is the fused sampler's; the histories re-run the steps one vita_flownet_forward_f32 call at a time on the same noise.
        zk = z.detach().float()
        traj_history, vel_history = [], []
        dt = 1.0 / num_steps
        for i in range(num_steps):
            num_t = i / num_steps * (1 - self.eps) + self.eps
            vt = ops.flownet_forward(model, zk, torch.full((zk.shape[0],), num_t, device=zk.device))
            sigma_t = self.sigma_t(num_t)
            if sigma_t > 0:
                pred_sigma = vt + (sigma_t ** 2) / (2 * (self.noise_scale ** 2) * ((1 - num_t) ** 2)) * \
                    (0.5 * num_t * (1 - num_t) * vt - 0.5 * (2 - num_t) * zk)
                zk = zk + pred_sigma * dt + sigma_t * float(np.sqrt(dt)) * noise[i].to(zk.device)
            else:
                zk = zk + vt * dt
            traj_history.append(zk.clone().cpu())
            vel_history.append(vt.clone().cpu())
        return out, (traj_history, vel_history)


FLOW_MATCHER_CLASSES = {                       # flow_matchers.py:137-145
    "conditional": ConditionalFlowMatcher,
    "exact": ExactOptimalTransportConditionalFlowMatcher,
    "target": TargetConditionalFlowMatcher,
    "schrodinger": SchrodingerBridgeConditionalFlowMatcher,
    "mean": MeanFlowConditionalFlowMatcher,
    "improved_mean": ImprovedMeanFlowConditionalFlowMatcher,
    "consistency": ConsistencyFlowMatcher,
}


def get_flow_matcher(name="conditional", **kwargs):
    """flow_matchers.py:148-154 (same ValueError text)."""
    try:
        cls = FLOW_MATCHER_CLASSES[name]
    except KeyError as exc:
        valid_names = ", ".join(sorted(FLOW_MATCHER_CLASSES))
        raise ValueError(f"Invalid flow matcher name: {name}. Expected one of: {valid_names}") from exc
    return cls(**kwargs)
