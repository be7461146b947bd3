"""A sampler time-out cannot be missed (csrc/vita_ode_persistent.hip TeamSync, csrc/vita_ode_resident.hip
GridSync): both in-launch samplers bound their hand-off spins; a workgroup whose wait runs out sets the status word and leaves.
With a tiny spin bound (vita_debug_set_spin_limit) the wait for the first hand-off always runs out: the outputs must then be NaN
(the poison pass enqueued behind the launch), the status word != 0, and VitaPolicy.select_action / validate must raise.  Run once:
the path is deterministic (the first hand-off takes tens of microseconds, the bound is two polls)."""
import pytest
import torch

from vita_amd import _lib as L
from vita_amd import ops
from vita_amd.synthetic import make_state_dict

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture
def tiny_spin_limit():
    ops.set_sampler_spin_limit(1)
    yield
    ops.set_sampler_spin_limit(0)


def _handles():
    sd = make_state_dict(533, 21, seed=0)
    return sd, ops.FlowNetHandle(sd, DEV), ops.DecoderHandle(sd, DEV)


@pytest.mark.parametrize("B", [4096, 25])      # the persistent launch, the weights-resident launch
def test_timeout_poisons_outputs_and_sets_status(B, tiny_spin_limit):
    sd, fw, dw = _handles()
    ow, ob = sd["obs_encoder.weight"].to(DEV).contiguous(), sd["obs_encoder.bias"].to(DEV).contiguous()
    import ctypes as C
    assert L.load().vita_ode_sample_takes_persistent(C.byref(fw.w), B) in (1, 2)      # an in-launch sampler takes this batch
    obs = torch.randn(B, 533, generator=torch.Generator().manual_seed(B)).to(DEV)
    acts, lat = ops.generate_actions(ow, ob, fw, dw, obs, 6, return_latents=True)
    st = ops.generate_actions_status(fw, dw, B, 6, DEV)
    assert st != 0
    assert torch.isnan(acts).all() and torch.isnan(lat).all()
    x0 = torch.randn(B, 512, generator=torch.Generator().manual_seed(1)).to(DEV)
    x, traj = ops.ode_sample(fw, x0, 3, return_traces=True)
    assert ops.ode_sample_status(fw, B, 3, DEV) != 0
    assert torch.isnan(x).all() and torch.isnan(traj).all()
    # the default bound back: the same buffers give a clean solve again
    ops.set_sampler_spin_limit(0)
    acts2 = ops.generate_actions(ow, ob, fw, dw, obs, 6)
    assert ops.generate_actions_status(fw, dw, B, 6, DEV) == 0
    assert torch.isfinite(acts2).all()
    ops.set_persistent_sampler(False)
    try:
        ref = ops.generate_actions(ow, ob, fw, dw, obs, 6)
    finally:
        ops.set_persistent_sampler(True)
    assert ((acts2 - ref).abs().max() / ref.abs().max()).item() <= 2e-6


def test_policy_raises_on_timeout(tiny_spin_limit):
    from vita_amd.config import default_config
    from vita_amd.policy import VitaPolicy
    cfg = default_config("hook_package")       # the default dims: 25 environments take the weights-resident launch
    g = torch.Generator().manual_seed(0)
    s, a = cfg.task.state_dim, cfg.task.action_dim
    stats = {cfg.task.state_key: {"mean": torch.randn(s, generator=g).tolist(), "std": (torch.rand(s, generator=g) + 0.5).tolist()},
             cfg.task.action_key: {"mean": torch.randn(a, generator=g).tolist(), "std": (torch.rand(a, generator=g) + 0.5).tolist()}}
    torch.manual_seed(0)
    policy = VitaPolicy(cfg, stats).to(DEV)
    batch = {cfg.task.state_key: torch.randn(25, s, generator=g).to(DEV), "observation.features": torch.relu(torch.randn(25, 512, generator=g)).to(DEV)}
    policy.reset()
    with pytest.raises(L.VitaHipError, match="timed out"):
        policy.select_action(batch)
    full = dict(batch)
    full = {k: v.unsqueeze(1) for k, v in full.items()}
    full[cfg.task.action_key] = torch.randn(25, cfg.policy.pred_horizon, a, generator=g).to(DEV)
    with pytest.raises(L.VitaHipError, match="timed out"):
        policy.validate(full)
    ops.set_sampler_spin_limit(0)
    policy.reset()
    act = policy.select_action(batch)
    assert act.shape == (25, a) and torch.isfinite(act).all()


def test_noalloc_status_and_poison(tiny_spin_limit):
    import ctypes as C
    lib = L.load()
    sd, fw, dw = _handles()
    ow, ob = sd["obs_encoder.weight"].to(DEV).contiguous(), sd["obs_encoder.bias"].to(DEV).contiguous()
    obs = torch.randn(25, 533, generator=torch.Generator().manual_seed(3)).to(DEV)
    got = torch.zeros(25, dw.OUT, device=DEV)
    rc = lib.vita_generate_actions_noalloc_f32(ow.data_ptr(), ob.data_ptr(), 533, C.byref(fw.w), C.byref(dw.w), obs.data_ptr(), 533,
                                               got.data_ptr(), dw.OUT, None, 25, 6, None)
    assert rc == 0, lib.vita_last_error()
    assert ops.generate_actions_noalloc_status(fw, 25, 6, DEV) != 0
    assert torch.isnan(got).all()
    ops.set_sampler_spin_limit(0)
    rc = lib.vita_generate_actions_noalloc_f32(ow.data_ptr(), ob.data_ptr(), 533, C.byref(fw.w), C.byref(dw.w), obs.data_ptr(), 533,
                                               got.data_ptr(), dw.OUT, None, 25, 6, None)
    assert rc == 0, lib.vita_last_error()
    assert ops.generate_actions_noalloc_status(fw, 25, 6, DEV) == 0
    assert torch.isfinite(got).all()
