"""GPU parity of the alternate matchers' inference paths through the C ABI: against the reference's
own outputs (tests/golden/alt.safetensors) and against the oracle in fp64 at the default dims.  Tolerance 1e-5 normalised
max error, as for the default sampler."""
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vita_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from vita_amd import _lib, ops as _ops
    _lib.load()
    return _ops


@pytest.fixture(scope="module")
def alt(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "alt.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_meanflow_golden(ops, dev, alt):
    sd = {k[len("mf/sd/"):]: v for k, v in alt.items() if k.startswith("mf/sd/")}
    mw = ops.MeanFlowNetHandle(sd, dev)
    assert (mw.D, mw.H, mw.FF, mw.L) == (32, 48, 192, 3)
    u, v = ops.meanflow_forward(mw, alt["mf/x"].to(dev), alt["mf/t"].to(dev), alt["mf/h"].to(dev))
    assert nerr(u, alt["mf/u"]) <= TOL and nerr(v, alt["mf/v"]) <= TOL
    u2, none = ops.meanflow_forward(mw, alt["mf/x"].to(dev), alt["mf/t"].to(dev), alt["mf/h"].to(dev), need_v=False)
    assert none is None and torch.equal(u, u2)
    for K in (1, 2, 5):
        xs = ops.meanflow_sample(mw, alt["mf/x"].to(dev), K)
        assert nerr(xs, alt[f"mf/sample/K{K}"]) <= TOL, K


def test_consistency_sampler_golden(ops, dev, alt):
    sd = {k[len("cfm/sd/"):]: v for k, v in alt.items() if k.startswith("cfm/sd/")}
    fw = ops.FlowNetHandle(sd, dev)
    x0 = alt["cfm/x0"].to(dev)
    for K in (1, 4):
        z = ops.consistency_sample(fw, x0, K, sigma_var=0.0)
        assert nerr(z, alt[f"cfm/det/K{K}"]) <= TOL, K
        z = ops.consistency_sample(fw, x0, K, noise=alt[f"cfm/sde/noise{K}"].to(dev))
        assert nerr(z, alt[f"cfm/sde/K{K}"]) <= TOL, K


@pytest.mark.parametrize("B,K", [(1, 1), (25, 1), (300, 3), (4096, 1)])
def test_meanflow_default_dims_vs_oracle_fp64(ops, dev, B, K):
    sd = O.make_meanflow_state_dict(512, 512, 4, 4.0, seed=8)
    g = torch.Generator().manual_seed(B + K)
    x = torch.randn(B, 512, generator=g)
    mw = ops.MeanFlowNetHandle(sd, dev)
    n = min(B, 64)                                   # oracle on a row sample (rows are independent)
    ref = O.mean_flow_sample({k: v.double() for k, v in sd.items()}, x[:n].double(), K)
    got = ops.meanflow_sample(mw, x.to(dev), K)
    assert torch.isfinite(got).all() and nerr(got[:n], ref) <= TOL
    if B > 1:                                        # per-row (t, h) entry point
        t, h = torch.rand(n, generator=g), torch.rand(n, generator=g)
        u64, v64, _ = O.mean_flow_net_forward({k: v.double() for k, v in sd.items()}, x[:n].double(), t.double(), h.double())
        u, v = ops.meanflow_forward(mw, x[:n].to(dev), t.to(dev), h.to(dev))
        assert nerr(u, u64) <= TOL and nerr(v, v64) <= TOL


@pytest.mark.parametrize("B,K,sigma_var", [(25, 1, 1.0), (512, 6, 1.0), (512, 6, 0.0), (4096, 2, 0.5)])
def test_consistency_sampler_default_dims_vs_oracle_fp64(ops, dev, B, K, sigma_var):
    sd = O.make_state_dict(533, 21, seed=2)
    g = torch.Generator().manual_seed(B * 3 + K)
    x = torch.randn(B, 512, generator=g)
    noise = torch.randn(K, B, 512, generator=g) if sigma_var > 0 else None
    fw = ops.FlowNetHandle(sd, dev)
    n = min(B, 32)
    sd64 = {k: v.double() for k, v in sd.items() if k.startswith("flow_net.")}
    ref = O.consistency_sample(sd64, x[:n].double(), K, sigma_var=sigma_var, noise=noise[:, :n].double() if noise is not None else None)
    got = ops.consistency_sample(fw, x.to(dev), K, sigma_var=sigma_var, noise=noise.to(dev) if noise is not None else None)
    assert torch.isfinite(got).all() and nerr(got[:n], ref) <= TOL


def test_alt_matchers_return_traces(ops, dev, alt):
    """return_traces of the two alternate samplers (mean_flow_matcher.py:217-233, consistency_flow_matcher.py:84-106): list
    lengths and contents as the reference builds them (MeanFlow: K+1 entries starting at x0 with a zero velocity; Consistency:
    K entries), the last history entry equal to the fused sampler's result within TOL, and x itself the fused result's bits."""
    from vita_amd import flow
    sd = {k[len("mf/sd/"):]: v for k, v in alt.items() if k.startswith("mf/sd/")}
    mw = ops.MeanFlowNetHandle(sd, dev)
    m = flow.get_flow_matcher("mean", num_sampling_steps=1)
    x0 = alt["mf/x"].to(dev)
    for K in (1, 2, 5):
        x, (traj, vel) = m.sample(mw, tuple(x0.shape), dev, num_steps=K, return_traces=True, start=x0)
        assert torch.equal(x, m.sample(mw, tuple(x0.shape), dev, num_steps=K, start=x0))
        assert nerr(x, alt[f"mf/sample/K{K}"]) <= TOL
        assert len(traj) == len(vel) == K + 1 and all(not h.is_cuda for h in traj + vel)
        assert torch.equal(traj[0], x0.cpu()) and not vel[0].any()
        assert nerr(traj[-1], alt[f"mf/sample/K{K}"]) <= TOL
        for k in range(K):                                  # x_{k+1} = x_k - u_k / K
            assert nerr(traj[k + 1], traj[k] - vel[k + 1] / K) <= 1e-6
    sd = {k[len("cfm/sd/"):]: v for k, v in alt.items() if k.startswith("cfm/sd/")}
    fw = ops.FlowNetHandle(sd, dev)
    x0 = alt["cfm/x0"].to(dev)
    for K in (1, 4):
        for sigma_var, key in ((0.0, "det"), (1.0, "sde")):
            c = flow.get_flow_matcher("consistency", num_sampling_steps=1, sigma_var=sigma_var)
            noise = alt[f"cfm/sde/noise{K}"].to(dev) if sigma_var > 0 else None
            z, (traj, vel) = c.sample(fw, tuple(x0.shape), dev, num_steps=K, return_traces=True, start=x0, noise=noise)
            assert torch.equal(z, c.sample(fw, tuple(x0.shape), dev, num_steps=K, start=x0, noise=noise))
            assert len(traj) == len(vel) == K and all(not h.is_cuda for h in traj + vel)
            assert nerr(traj[-1], alt[f"cfm/{key}/K{K}"]) <= TOL and nerr(z, alt[f"cfm/{key}/K{K}"]) <= TOL
            v0 = ops.flownet_forward(fw, x0, torch.full((x0.shape[0],), c.eps, device=dev))
            assert torch.equal(vel[0], v0.cpu())


def test_alt_sampler_errors(ops, dev):
    from vita_amd._lib import VitaHipError
    sd = O.make_state_dict(20, 7, latent_dim=32, hidden_dim=32, num_layers=2, ae_hidden=32, ae_layers=2, time_embed_dim=64, seed=4)
    fw = ops.FlowNetHandle(sd, dev)
    x = torch.zeros(4, 32, device=dev)
    with pytest.raises(VitaHipError):
        ops.consistency_sample(fw, x, 2)                                   # sigma_var = 1.0 without noise
    with pytest.raises(VitaHipError):
        ops.consistency_sample(fw, x, 0, sigma_var=0.0)
    with pytest.raises(VitaHipError):
        ops.consistency_sample(fw, x, 2, noise=torch.zeros(3, 4, 32, device=dev))
    mw = ops.MeanFlowNetHandle(O.make_meanflow_state_dict(32, 48, 3, 4.0), dev)
    with pytest.raises(VitaHipError):
        ops.meanflow_sample(mw, torch.zeros(4, 33, device=dev), 1)
    with pytest.raises(VitaHipError):
        ops.meanflow_sample(mw, torch.zeros(4, 32, device=dev), 0)
