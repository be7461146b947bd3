"""The persistent Euler sampler (csrc/vita_ode_persistent.hip: all K steps of TorchFlowMatcher.sample, flow_matchers.py:41-79,
in one launch whose workgroups hand activations to each other) against the per-launch path it replaces at chip-filling
batches: same tiles and epilogues, so the SAME BITS; and against the fp64 oracle on sampled rows."""
import pytest
import torch

from oracle import vita_oracle as O
from vita_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_switch():
    prev = ops.set_persistent_sampler(True)
    yield
    ops.set_persistent_sampler(prev)


def both_paths(fw, x0, K, traces=False):
    ops.set_persistent_sampler(True)
    a = ops.ode_sample(fw, x0, K, return_traces=traces)
    assert ops.ode_sample_status(fw, x0.shape[0], K, DEV) == 0
    ops.set_persistent_sampler(False)
    b = ops.ode_sample(fw, x0, K, return_traces=traces)
    torch.cuda.synchronize()
    ops.set_persistent_sampler(True)
    return a, b


@pytest.mark.parametrize("B,K", [(4096, 6), (4096, 1), (8192, 3)])
def test_persistent_equals_per_launch_path_bitwise(B, K):
    sd = O.make_state_dict(533, 21, seed=0)
    fw = ops.FlowNetHandle(sd, DEV)
    x0 = torch.randn(B, 512, generator=torch.Generator().manual_seed(B + K)).to(DEV)
    a, b = both_paths(fw, x0, K)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), (a - b).abs().max().item()
    # and it is the reference's arithmetic: sampled rows against the fp64 oracle (tolerance of the headline path: 1e-5)
    idx = torch.arange(0, B, B // 8)
    ref = O.euler_sample({k: v.double() for k, v in sd.items()}, x0[idx].cpu().double(), K)
    err = ((a[idx].cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 1e-5, err


def test_persistent_trajectory_and_repeatability():
    """return_traces (flow_matchers.py:63-66,73-75): every x_k of the trajectory, bitwise; ten solves in a row on the same
    buffers (consumers' L1 warm with the previous solve's lines) give the same bits every time."""
    sd = O.make_state_dict(555, 7, seed=3)
    fw = ops.FlowNetHandle(sd, DEV)
    x0 = torch.randn(4096, 512, generator=torch.Generator().manual_seed(11)).to(DEV)
    (a, ta), (b, tb) = both_paths(fw, x0, 4, traces=True)
    assert torch.equal(ta, tb) and torch.equal(a, b) and torch.equal(ta[0], x0) and torch.equal(ta[-1], a)
    first = ops.ode_sample(fw, x0, 6).clone()
    for _ in range(10):
        again = ops.ode_sample(fw, x0, 6)
        assert torch.equal(again, first)
    assert ops.ode_sample_status(fw, 4096, 6, DEV) == 0


def test_persistent_under_uneven_load():
    """Hand-offs must not depend on timing or placement: the same solve while another stream keeps part of the chip busy
    with unrelated streaming kernels (workgroups start late, teams drift apart) gives the same bits."""
    sd = O.make_state_dict(533, 21, seed=0)
    fw = ops.FlowNetHandle(sd, DEV)
    x0 = torch.randn(4096, 512, generator=torch.Generator().manual_seed(5)).to(DEV)
    want = ops.ode_sample(fw, x0, 6).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(DEV)
    junk = torch.randn(64 << 20, device=DEV)
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(40):
                junk.mul_(1.0001)
        got = ops.ode_sample(fw, x0, 6)
        torch.cuda.synchronize()
        assert torch.equal(got, want), rep
    assert ops.ode_sample_status(fw, 4096, 6, DEV) == 0


def test_other_team_size_and_two_tiles_per_member():
    """hidden_dim 256: teams of 8 workgroups, 32 row groups per launch, and out_proj has 16 tiles per row group, two per
    member (the tile loop inside a phase).  (hidden_dim 1024 would need teams of 32: measured slower than the per-launch
    path, so that shape stays on it — both switches then run the same code.)"""
    sd = O.make_state_dict(533, 21, hidden_dim=256, seed=1)
    fw = ops.FlowNetHandle(sd, DEV)
    x0 = torch.randn(8192, 512, generator=torch.Generator().manual_seed(6)).to(DEV)
    a, b = both_paths(fw, x0, 2)
    assert torch.equal(a, b)
    idx = torch.arange(0, 8192, 1024)
    ref = O.euler_sample({k: v.double() for k, v in sd.items()}, x0[idx].cpu().double(), 2)
    assert ((a[idx].cpu().double() - ref).abs().max() / ref.abs().max()).item() <= 1e-5
    sdw = O.make_state_dict(533, 21, hidden_dim=1024, seed=1)
    fww = ops.FlowNetHandle(sdw, DEV)
    xw = x0[:2048].contiguous()
    aw, bw = both_paths(fww, xw, 2)
    assert torch.equal(aw, bw)


def test_generate_actions_uses_it_and_matches():
    sd = O.make_state_dict(533, 21, seed=0)
    fw, dw = ops.FlowNetHandle(sd, DEV), ops.DecoderHandle(sd, DEV)
    ow, ob = sd["obs_encoder.weight"].to(DEV), sd["obs_encoder.bias"].to(DEV)
    obs = torch.randn(4096, 533, generator=torch.Generator().manual_seed(7)).to(DEV)
    ops.set_persistent_sampler(True)
    a = ops.generate_actions(ow, ob, fw, dw, obs, 6)
    ops.set_persistent_sampler(False)
    b = ops.generate_actions(ow, ob, fw, dw, obs, 6)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_k100_robomimic_square_b4096_against_fp64_oracle():
    """BASELINE configs[3] at its headline shape through the persistent launch: robomimic_square dims, 4096 rows, K = 100 (600
    phases, team counters to 9 600) against the fp64 oracle on 16 sampled rows, and the status word."""
    sd = O.make_state_dict(555, 7, seed=0)
    fw = ops.FlowNetHandle(sd, DEV)
    x0 = torch.randn(4096, 512, generator=torch.Generator().manual_seed(100)).to(DEV)
    a = ops.ode_sample(fw, x0, 100)
    assert ops.ode_sample_status(fw, 4096, 100, DEV) == 0
    assert torch.isfinite(a).all()
    idx = torch.arange(0, 4096, 256) + torch.arange(16) * 13
    ref = O.euler_sample({k: v.double() for k, v in sd.items()}, x0[idx].cpu().double(), 100)
    err = ((a[idx].cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 1e-5, err


def test_wide_net_h2048_b4096_against_fp64_oracle_and_position_independence():
    """BASELINE configs[4] at its headline shape: hidden_dim 2048 (FF = 8192), 4096 rows, K = 6 (the per-launch path with the
    folded LayerNorms: teams of 64 would not fit the persistent launch) against the fp64 oracle on sampled rows; and a row's
    result does not depend on where in the batch it sits, bit for bit."""
    sd = O.make_state_dict(533, 21, hidden_dim=2048, seed=3)
    fw = ops.FlowNetHandle(sd, DEV)
    x0 = torch.randn(4096, 512, generator=torch.Generator().manual_seed(2048)).to(DEV)
    a = ops.ode_sample(fw, x0, 6)
    assert torch.isfinite(a).all()
    idx = torch.tensor([0, 63, 64, 255, 256, 1000, 2047, 2048, 3333, 4095])
    ref = O.euler_sample({k: v.double() for k, v in sd.items()}, x0[idx].cpu().double(), 6)
    err = ((a[idx].cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 1e-5, err
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(5)).to(DEV)
    b = ops.ode_sample(fw, x0[perm].contiguous(), 6)
    assert torch.equal(b, a[perm])
