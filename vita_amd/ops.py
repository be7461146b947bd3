"""Thin functional layer: torch CUDA(HIP) tensors in, C-ABI calls out.

PyTorch is used for device memory, streams and shapes only; every FLOP on this path runs in
libvita_hip.so.  Nothing here falls back to torch math.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import torch

from . import _lib as L

Tensor = torch.Tensor


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32(t: Tensor, what: str) -> Tensor:
    if not t.is_cuda:
        raise L.VitaHipError(f"{what}: expected a tensor on the GPU, got device {t.device} "
                             "(vita_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise L.VitaHipError(f"{what}: expected float32, got {t.dtype}")
    return t


def _rows(t: Tensor, what: str) -> Tensor:
    """2-D view with unit inner stride (row stride free)."""
    _f32(t, what)
    if t.dim() != 2:
        raise L.VitaHipError(f"{what}: expected a 2-D tensor, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


_workspaces: dict = {}


def workspace(device: torch.device, nbytes: int) -> Tensor:
    """Per-(device, stream) scratch buffer, grown on demand; the C side never allocates."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# --------------------------------------------------------------------------------------
# fused linear
# --------------------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None, act: int = L.ACT_NONE, *,
           pro: int = L.PRO_NONE, ln_eps: float = 0.0, pro_mul: Optional[Tensor] = None,
           pro_add: Optional[Tensor] = None, gate: Optional[Tensor] = None,
           gate_scalar: Optional[float] = None, resid: Optional[Tensor] = None,
           out: Optional[Tensor] = None) -> Tensor:
    """Y = resid + gate * act(pro(x) @ w.T + b) — see vita_linear_f32 in include/vita_hip.h."""
    lib = L.load()
    x = _rows(x, "linear: x")
    w = _rows(w, "linear: w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise L.VitaHipError(f"linear: x is (M,{K}) but w is ({N},{w.shape[1]})")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    out = _f32(out, "linear: out")
    a = L.LinearArgs()
    a.X, a.ldx = x.data_ptr(), _ld(x)
    a.W, a.ldw = w.data_ptr(), _ld(w)
    a.bias = _f32(b, "linear: b").contiguous().data_ptr() if b is not None else None
    a.Y, a.ldy = out.data_ptr(), _ld(out)
    a.M, a.N, a.K = M, N, K
    a.act, a.pro, a.ln_eps = act, pro, ln_eps
    keep = []
    for name, t in (("pro_mul", pro_mul), ("pro_add", pro_add), ("gate", gate)):
        if t is None:
            continue
        _f32(t, f"linear: {name}")
        if t.dim() == 1:
            t = t.contiguous()
            setattr(a, name, t.data_ptr())
            setattr(a, "ld_" + name, 0)
        else:
            t = _rows(t, f"linear: {name}")
            setattr(a, name, t.data_ptr())
            setattr(a, "ld_" + name, _ld(t))
        keep.append(t)
    if gate_scalar is not None:
        a.use_gate_scalar, a.gate_scalar = 1, float(gate_scalar)
    if resid is not None:
        resid = _rows(resid, "linear: resid")
        a.resid, a.ld_resid = resid.data_ptr(), _ld(resid)
    L.check(lib.vita_linear_f32(C.byref(a), _stream(x)), "vita_linear_f32")
    return out


# --------------------------------------------------------------------------------------
# weight descriptors
# --------------------------------------------------------------------------------------
def _prep(sd: Mapping[str, Tensor], key: str, device) -> Tensor:
    t = sd[key]
    t = t.detach()
    if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


class FlowNetHandle:
    """Device pointers of a SimpleFlowNet state dict (keys as in simple_flow_net.py:52-84,
    prefix 'flow_net.' by default).  Holds references so the tensors outlive the calls."""

    def __init__(self, sd: Mapping[str, Tensor], device, prefix: str = "flow_net."):
        device = torch.device(device)
        self.device = device
        self.t = {}
        g = lambda k: self.t.setdefault(k, _prep(sd, prefix + k, device))  # noqa: E731
        w = L.FlowNetWeights()
        in_w = g("input_proj.weight")
        w.H, w.D = in_w.shape
        w.TE = g("time_embed.1.weight").shape[1]
        n = 0
        while f"{prefix}layers.{n}.mlp.fc1.weight" in sd:
            n += 1
        if n > L.MAX_LAYERS:
            raise L.VitaHipError(f"flow net has {n} layers; the ABI supports {L.MAX_LAYERS}")
        w.L = n
        w.FF = g("layers.0.mlp.fc1.weight").shape[0] if n else w.H
        w.in_w, w.in_b = in_w.data_ptr(), g("input_proj.bias").data_ptr()
        w.te1_w, w.te1_b = g("time_embed.1.weight").data_ptr(), g("time_embed.1.bias").data_ptr()
        w.te3_w, w.te3_b = g("time_embed.3.weight").data_ptr(), g("time_embed.3.bias").data_ptr()
        for l in range(n):
            w.fc1_w[l] = g(f"layers.{l}.mlp.fc1.weight").data_ptr()
            w.fc1_b[l] = g(f"layers.{l}.mlp.fc1.bias").data_ptr()
            w.fc2_w[l] = g(f"layers.{l}.mlp.fc2.weight").data_ptr()
            w.fc2_b[l] = g(f"layers.{l}.mlp.fc2.bias").data_ptr()
            w.mod_w[l] = g(f"layers.{l}.time_modulator.1.weight").data_ptr()
            w.mod_b[l] = g(f"layers.{l}.time_modulator.1.bias").data_ptr()
        w.norm_w, w.norm_b = g("norm.weight").data_ptr(), g("norm.bias").data_ptr()
        w.out_w, w.out_b = g("out_proj.weight").data_ptr(), g("out_proj.bias").data_ptr()
        if g("out_proj.weight").shape != (w.D, w.H):
            raise L.VitaHipError("flow net: out_proj shape does not match (D,H)")
        self.w = w
        self.D, self.H, self.FF, self.L = w.D, w.H, w.FF, w.L


class DecoderHandle:
    """Device pointers of a SimpleActionDecoder state dict (action_ae.py:226-258)."""

    def __init__(self, sd: Mapping[str, Tensor], device, prefix: str = "action_decoder."):
        device = torch.device(device)
        self.device = device
        self.t = {}
        g = lambda k: self.t.setdefault(k, _prep(sd, prefix + k, device))  # noqa: E731
        w = L.DecoderWeights()
        w.HD, w.D = g("input_proj.weight").shape
        w.OUT = g("output_proj.weight").shape[0]
        n = 0
        while f"{prefix}layers.{n}.fc1.weight" in sd:
            n += 1
        if n > L.MAX_LAYERS:
            raise L.VitaHipError(f"decoder has {n} layers; the ABI supports {L.MAX_LAYERS}")
        w.NL = n
        w.in_w, w.in_b = g("input_proj.weight").data_ptr(), g("input_proj.bias").data_ptr()
        for i in range(n):
            w.fc1_w[i] = g(f"layers.{i}.fc1.weight").data_ptr()
            w.fc1_b[i] = g(f"layers.{i}.fc1.bias").data_ptr()
            w.fc2_w[i] = g(f"layers.{i}.fc2.weight").data_ptr()
            w.fc2_b[i] = g(f"layers.{i}.fc2.bias").data_ptr()
        w.out_w, w.out_b = g("output_proj.weight").data_ptr(), g("output_proj.bias").data_ptr()
        self.w = w
        self.D, self.HD, self.OUT = w.D, w.HD, w.OUT


# --------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------
def flownet_forward(fw: FlowNetHandle, x: Tensor, t: Tensor) -> Tensor:
    """SimpleFlowNet.forward(x, t), per-row t (simple_flow_net.py:99-108)."""
    lib = L.load()
    x = _rows(x, "flownet_forward: x")
    t = _f32(t, "flownet_forward: t").contiguous()
    B = x.shape[0]
    if x.shape[1] != fw.D or t.shape != (B,):
        raise L.VitaHipError(f"flownet_forward: x {tuple(x.shape)} / t {tuple(t.shape)} do not match D={fw.D}")
    v = torch.empty((B, fw.D), dtype=torch.float32, device=x.device)
    nbytes = lib.vita_flownet_forward_workspace_bytes(C.byref(fw.w), B)
    ws = workspace(x.device, nbytes)
    L.check(lib.vita_flownet_forward_f32(C.byref(fw.w), x.data_ptr(), _ld(x), t.data_ptr(), v.data_ptr(), fw.D, B,
                                         ws.data_ptr(), ws.numel(), _stream(x)), "vita_flownet_forward_f32")
    return v


def ode_sample(fw: FlowNetHandle, x0: Tensor, num_steps: int, return_traces: bool = False):
    """TorchFlowMatcher.sample(start=x0) (flow_matchers.py:55-79)."""
    lib = L.load()
    x0 = _rows(x0, "ode_sample: x0")
    B = x0.shape[0]
    if x0.shape[1] != fw.D:
        raise L.VitaHipError(f"ode_sample: x0 has {x0.shape[1]} columns, flow net D={fw.D}")
    out = torch.empty((B, fw.D), dtype=torch.float32, device=x0.device)
    traj = torch.empty((num_steps + 1, B, fw.D), dtype=torch.float32, device=x0.device) if return_traces else None
    nbytes = lib.vita_ode_sample_workspace_bytes(C.byref(fw.w), B, num_steps)
    ws = workspace(x0.device, nbytes)
    L.check(lib.vita_ode_sample_f32(C.byref(fw.w), x0.data_ptr(), _ld(x0), out.data_ptr(), fw.D, B, num_steps,
                                    traj.data_ptr() if traj is not None else None,
                                    ws.data_ptr(), ws.numel(), _stream(x0)), "vita_ode_sample_f32")
    return (out, traj) if return_traces else out


def decoder_forward(dw: DecoderHandle, z: Tensor) -> Tensor:
    """SimpleActionDecoder.forward (action_ae.py:267-279); returns (B, OUT)."""
    lib = L.load()
    z = _rows(z, "decoder_forward: z")
    B = z.shape[0]
    if z.shape[1] != dw.D:
        raise L.VitaHipError(f"decoder_forward: z has {z.shape[1]} columns, decoder D={dw.D}")
    out = torch.empty((B, dw.OUT), dtype=torch.float32, device=z.device)
    nbytes = lib.vita_decoder_forward_workspace_bytes(C.byref(dw.w), B)
    ws = workspace(z.device, nbytes)
    L.check(lib.vita_decoder_forward_f32(C.byref(dw.w), z.data_ptr(), _ld(z), out.data_ptr(), dw.OUT, B,
                                         ws.data_ptr(), ws.numel(), _stream(z)), "vita_decoder_forward_f32")
    return out


def generate_actions(obs_w: Tensor, obs_b: Tensor, fw: FlowNetHandle, dw: DecoderHandle, obs_features: Tensor,
                     num_steps: int, return_latents: bool = False, out: Optional[Tensor] = None):
    """VitaPolicy.generate_actions from observer features on (vita_policy.py:191-208)."""
    lib = L.load()
    obs_features = _rows(obs_features, "generate_actions: obs_features")
    obs_w = _rows(obs_w, "generate_actions: obs_w")
    obs_b = _f32(obs_b, "generate_actions: obs_b").contiguous()
    B, O = obs_features.shape
    if obs_w.shape != (fw.D, O) or not obs_w.is_contiguous():
        if obs_w.shape != (fw.D, O):
            raise L.VitaHipError(f"generate_actions: obs_encoder weight {tuple(obs_w.shape)} != ({fw.D},{O})")
        obs_w = obs_w.contiguous()
    if out is None:
        out = torch.empty((B, dw.OUT), dtype=torch.float32, device=obs_features.device)
    lat = torch.empty((B, fw.D), dtype=torch.float32, device=obs_features.device) if return_latents else None
    nbytes = lib.vita_generate_actions_workspace_bytes(C.byref(fw.w), C.byref(dw.w), B, num_steps)
    ws = workspace(obs_features.device, nbytes)
    L.check(lib.vita_generate_actions_f32(obs_w.data_ptr(), obs_b.data_ptr(), O, C.byref(fw.w), C.byref(dw.w),
                                          obs_features.data_ptr(), _ld(obs_features), out.data_ptr(), dw.OUT,
                                          lat.data_ptr() if lat is not None else None, B, num_steps,
                                          ws.data_ptr(), ws.numel(), _stream(obs_features)),
            "vita_generate_actions_f32")
    return (out, lat) if return_latents else out


# --------------------------------------------------------------------------------------
# losses / elementwise
# --------------------------------------------------------------------------------------
def _loss_fwd(fn_name: str, a: Tensor, b: Tensor) -> Tensor:
    lib = L.load()
    a = _f32(a, fn_name).contiguous()
    b = _f32(b, fn_name).contiguous()
    if a.shape != b.shape:
        raise L.VitaHipError(f"{fn_name}: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    part = torch.empty(lib.vita_loss_partials_count(), dtype=torch.float32, device=a.device)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    L.check(getattr(lib, fn_name)(a.data_ptr(), b.data_ptr(), a.numel(), part.data_ptr(), out.data_ptr(), _stream(a)),
            fn_name)
    return out


def _loss_bwd(fn_name: str, a: Tensor, b: Tensor, gout: Tensor, need_a=True, need_b=True):
    lib = L.load()
    a = _f32(a, fn_name).contiguous()
    b = _f32(b, fn_name).contiguous()
    gout = _f32(gout, fn_name).reshape(1).contiguous()
    ga = torch.empty_like(a) if need_a else None
    gb = torch.empty_like(b) if need_b else None
    L.check(getattr(lib, fn_name)(a.data_ptr(), b.data_ptr(), a.numel(), gout.data_ptr(),
                                  ga.data_ptr() if ga is not None else None,
                                  gb.data_ptr() if gb is not None else None, _stream(a)), fn_name)
    return ga, gb


def mse_loss_fwd(a, b): return _loss_fwd("vita_mse_loss_fwd_f32", a, b)
def l1_loss_fwd(a, b): return _loss_fwd("vita_l1_loss_fwd_f32", a, b)
def mse_loss_bwd(a, b, g, need_a=True, need_b=True): return _loss_bwd("vita_mse_loss_bwd_f32", a, b, g, need_a, need_b)
def l1_loss_bwd(a, b, g, need_a=True, need_b=True): return _loss_bwd("vita_l1_loss_bwd_f32", a, b, g, need_a, need_b)


def _idx(t, device):
    return None if t is None else t.to(device=device, dtype=torch.int64).contiguous()


def cfm_interpolate(x0: Tensor, x1: Tensor, t: Tensor, pair_j: Optional[Tensor] = None, pair_i: Optional[Tensor] = None):
    """xt = t*x1[pair_j] + (1-t)*x0[pair_i], ut = x1[pair_j] - x0[pair_i] (torchcfm, sigma=0; flow_matchers.py:34)."""
    lib = L.load()
    x0 = _f32(x0, "cfm_interpolate").contiguous()
    x1 = _f32(x1, "cfm_interpolate").contiguous()
    t = _f32(t, "cfm_interpolate").contiguous()
    B, D = x0.shape
    pair_i, pair_j = _idx(pair_i, x0.device), _idx(pair_j, x0.device)
    xt = torch.empty_like(x0)
    ut = torch.empty_like(x0)
    L.check(lib.vita_cfm_interpolate_f32(x0.data_ptr(), x1.data_ptr(), t.data_ptr(),
                                         pair_i.data_ptr() if pair_i is not None else None,
                                         pair_j.data_ptr() if pair_j is not None else None,
                                         xt.data_ptr(), ut.data_ptr(), B, D, _stream(x0)), "vita_cfm_interpolate_f32")
    return xt, ut


def _norm(fn_name: str, x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    lib = L.load()
    x = _f32(x, fn_name).contiguous()
    mean = _f32(mean, fn_name).contiguous().reshape(-1)
    std = _f32(std, fn_name).contiguous().reshape(-1)
    Cdim = mean.numel()
    if x.shape[-1] != Cdim or std.numel() != Cdim:
        raise L.VitaHipError(f"{fn_name}: last dim {x.shape[-1]} != stats dim {Cdim}")
    y = torch.empty_like(x)
    L.check(getattr(lib, fn_name)(x.data_ptr(), mean.data_ptr(), std.data_ptr(), y.data_ptr(), x.numel() // Cdim, Cdim,
                                  _stream(x)), fn_name)
    return y


def normalize(x, mean, std): return _norm("vita_normalize_f32", x, mean, std)
def unnormalize(x, mean, std): return _norm("vita_unnormalize_f32", x, mean, std)


def layernorm_modulate(x: Tensor, mul: Tensor, add: Tensor, eps: float, mode: int = L.PRO_LN_MOD) -> Tensor:
    """LN(x)*(1+mul)+add (PRO_LN_MOD, simple_flow_net.py:44-45) or LN(x)*mul+add (PRO_LN_AFFINE, :81,106)."""
    lib = L.load()
    x = _rows(x, "layernorm_modulate: x")
    M, H = x.shape
    y = torch.empty((M, H), dtype=torch.float32, device=x.device)
    args = []
    for name, t in (("mul", mul), ("add", add)):
        _f32(t, "layernorm_modulate: " + name)
        if t.dim() == 1:
            t = t.contiguous()
            args.append((t, 0))
        else:
            t = _rows(t, "layernorm_modulate: " + name)
            args.append((t, _ld(t)))
    L.check(lib.vita_layernorm_modulate_f32(x.data_ptr(), _ld(x), y.data_ptr(), H, M, H, eps, mode,
                                            args[0][0].data_ptr(), args[0][1], args[1][0].data_ptr(), args[1][1],
                                            _stream(x)), "vita_layernorm_modulate_f32")
    return y


def sinusoidal_embed(t: Tensor, dim: int) -> Tensor:
    lib = L.load()
    t = _f32(t, "sinusoidal_embed").contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    L.check(lib.vita_sinusoidal_embed_f32(t.data_ptr(), out.data_ptr(), dim, t.numel(), dim, _stream(t)),
            "vita_sinusoidal_embed_f32")
    return out
