"""The allocation-free entry point (vita_generate_actions_noalloc_f32, include/vita_hip.h) against the workspace-taking
vita_generate_actions_f32 it is a variant of: two calls in a row through the C ABI, the second with a smaller input, so
that whatever the first call left in the shared scratch array is there when the second runs."""
import ctypes as C

import pytest
import torch

from vita_amd import _lib, ops

pytestmark = pytest.mark.gpu
TOL = 1e-6   # normalised max error; fp32 GEMMs whose split-K plan may differ (smaller split-K region): summation order only


def _weights(dev):
    from vita_amd.synthetic import make_state_dict
    sd = make_state_dict(533, 21, seed=0)
    return sd, ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)


def test_noalloc_matches_workspace_entry_point_twice():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    assert lib.vita_noalloc_init() == 0, lib.vita_last_error()
    sd, fw, dw = _weights(dev)
    ow, ob = sd["obs_encoder.weight"].to(dev).contiguous(), sd["obs_encoder.bias"].to(dev).contiguous()
    g = torch.Generator().manual_seed(7)
    for B, K in ((200, 6), (25, 3)):          # the second call is smaller and runs on the first call's leftovers
        obs = torch.randn(B, 533, generator=g).to(dev)
        want, want_lat = ops.generate_actions(ow, ob, fw, dw, obs, K, return_latents=True)
        got = torch.full((B, dw.OUT), float("nan"), device=dev)
        lat = torch.full((B, fw.D), float("nan"), device=dev)
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(dev)
        rc = lib.vita_generate_actions_noalloc_f32(ow.data_ptr(), ob.data_ptr(), 533, C.byref(fw.w), C.byref(dw.w), obs.data_ptr(), 533,
                                                   got.data_ptr(), dw.OUT, lat.data_ptr(), B, K, None)
        assert rc == 0, lib.vita_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) == before
        for a, b in ((got, want), (lat, want_lat)):
            assert torch.isfinite(a).all()
            err = ((a - b).abs().max() / b.abs().max()).item()
            assert err <= TOL, (B, K, err)


def test_noalloc_rejects_sizes_beyond_its_array():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    sd, fw, dw = _weights(dev)
    cap = lib.vita_noalloc_scratch_bytes()
    assert cap == 16 << 20
    assert lib.vita_generate_actions_noalloc_need_bytes(C.byref(fw.w), C.byref(dw.w), 200, 6) <= cap
    assert lib.vita_generate_actions_noalloc_need_bytes(C.byref(fw.w), C.byref(dw.w), 4096, 6) > cap
    x = torch.zeros(4096, 533, device=dev)
    y = torch.zeros(4096, dw.OUT, device=dev)
    ow, ob = sd["obs_encoder.weight"].to(dev).contiguous(), sd["obs_encoder.bias"].to(dev).contiguous()
    rc = lib.vita_generate_actions_noalloc_f32(ow.data_ptr(), ob.data_ptr(), 533, C.byref(fw.w), C.byref(dw.w), x.data_ptr(), 533,
                                               y.data_ptr(), dw.OUT, None, 4096, 6, None)
    assert rc == _lib.load().vita_generate_actions_noalloc_f32.restype(-3) or rc == -3
    assert b"scratch" in lib.vita_last_error()
