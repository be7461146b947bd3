"""Host-side mirror of flare/flow/flow_matchers.py for the torchcfm-backed matchers.

Same names, constructor arguments and error behaviour as the reference (`get_flow_matcher(name=..., sigma=...,
num_sampling_steps=...)`, `FLOW_MATCHER_CLASSES`, `.sample(model, shape, device, num_steps, return_traces,
start)`, `.compute_loss(model, target, start)`), but `model` is a `vita_amd.ops.FlowNetHandle` and the arithmetic
runs in libvita_hip.so: `.sample` is ONE C-ABI call that enqueues all K Euler steps.

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
    """flow_matchers.py:10-79."""

    pairing = "identity"

    def __init__(self, num_sampling_steps: int = 6, sigma: float = 0.0, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected flow matcher arguments: {sorted(kwargs)}")
        if float(sigma) != 0.0:
            raise NotImplementedError("vita_amd implements the sigma=0 path (vita.yaml:24); sigma > 0 adds sigma*eps to xt")
        self.sigma = float(sigma)
        self.num_sampling_steps = num_sampling_steps

    # -- the draw (host) ---------------------------------------------------------------
    def draw(self, x0: torch.Tensor, x1: torch.Tensor):
        """(t, pair_i, pair_j) for one compute_loss call; pair_* are None for the identity pairing."""
        pi = pj = None
        if self.pairing == "exact":
            pi, pj = exact_ot_pairing(x0.detach(), x1.detach())
        return draw_t(x0.shape[0], x0), pi, pj

    # -- reference API -----------------------------------------------------------------
    def compute_loss(self, model: "ops.FlowNetHandle", target: torch.Tensor, start: Optional[torch.Tensor] = None, **kwargs):
        """flow_matchers.py:19-39 (forward only; training uses the fused vita_compute_loss_f32)."""
        x0 = torch.randn_like(target) if start is None else start
        t, pi, pj = self.draw(x0, target)
        xt, ut = ops.cfm_interpolate(x0, target, t, pair_j=pj, pair_i=pi)
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


def _not_yet(name):
    class _NotBuilt:
        def __init__(self, *a, **k):
            raise NotImplementedError(
                f"flow matcher '{name}' is outside this build's hot path (SURVEY.md 8(f)); use 'exact' or 'conditional'")
    _NotBuilt.__name__ = name
    return _NotBuilt


FLOW_MATCHER_CLASSES = {                       # flow_matchers.py:137-145
    "conditional": ConditionalFlowMatcher,
    "exact": ExactOptimalTransportConditionalFlowMatcher,
    "target": _not_yet("target"),
    "schrodinger": _not_yet("schrodinger"),
    "mean": _not_yet("mean"),
    "improved_mean": _not_yet("improved_mean"),
    "consistency": _not_yet("consistency"),
}


def get_flow_matcher(name="conditional", **kwargs):
    """flow_matchers.py:148-154 (same ValueError text)."""
    try:
        cls = FLOW_MATCHER_CLASSES[name]
    except KeyError as exc:
        valid_names = ", ".join(sorted(FLOW_MATCHER_CLASSES))
        raise ValueError(f"Invalid flow matcher name: {name}. Expected one of: {valid_names}") from exc
    return cls(**kwargs)
