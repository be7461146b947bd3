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


_splitk: dict = {}


def _splitk_ws(device: torch.device, a) -> Tensor:
    """Per-(device, stream) split-K scratch for single vita_linear_f32 calls; returned so the caller keeps it alive."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _splitk.get(key)
    if buf is None:
        buf = torch.empty(L.load().vita_splitk_workspace_bytes(), dtype=torch.uint8, device=device)
        _splitk[key] = buf
    a.splitk_ws, a.splitk_ws_bytes = buf.data_ptr(), buf.numel()
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
    _splitk_ws(x.device, a)
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


def ode_sample_status(fw: FlowNetHandle, B: int, num_steps: int, device) -> int:
    """vita_ode_sample_status for the cached workspace the last ode_sample on `device` used (synchronises): 0 = the
    persistent launch completed, != 0 = a bounded hand-off wait ran out in phase status - 1."""
    lib = L.load()
    torch.cuda.synchronize(device)
    nbytes = lib.vita_ode_sample_workspace_bytes(C.byref(fw.w), B, num_steps)
    ws = workspace(torch.device(device), nbytes)
    st = C.c_int32(0)
    L.check(lib.vita_ode_sample_status(C.byref(fw.w), B, num_steps, ws.data_ptr(), C.byref(st)), "vita_ode_sample_status")
    return int(st.value)


def generate_actions_status(fw: FlowNetHandle, dw: "DecoderHandle", B: int, num_steps: int, device) -> int:
    """vita_generate_actions_status for the cached workspace the last generate_actions on `device` used (synchronises):
    0 = its in-launch sampler completed (or it ran launch by launch), != 0 = a bounded hand-off wait ran out and the actions
    are not valid."""
    lib = L.load()
    torch.cuda.synchronize(device)
    nbytes = lib.vita_generate_actions_workspace_bytes(C.byref(fw.w), C.byref(dw.w), B, num_steps)
    ws = workspace(torch.device(device), nbytes)
    st = C.c_int32(0)
    L.check(lib.vita_generate_actions_status(C.byref(fw.w), B, num_steps, ws.data_ptr(), C.byref(st)), "vita_generate_actions_status")
    return int(st.value)


def generate_actions_noalloc_status(fw: FlowNetHandle, B: int, num_steps: int, device) -> int:
    """vita_generate_actions_noalloc_status: the status word of the last generate_actions_noalloc with these shapes on
    `device` (synchronises)."""
    lib = L.load()
    torch.cuda.synchronize(device)
    st = C.c_int32(0)
    with torch.cuda.device(device):
        L.check(lib.vita_generate_actions_noalloc_status(C.byref(fw.w), B, num_steps, C.byref(st)), "vita_generate_actions_noalloc_status")
    return int(st.value)


def generate_actions_bf16_status(fw: FlowNetHandle, dw: "DecoderHandle", O: int, B: int, num_steps: int, device) -> int:
    """vita_generate_actions_bf16_status for the cached workspace the last generate_actions_bf16 on `device` used (synchronises)."""
    lib = L.load()
    torch.cuda.synchronize(device)
    nbytes = lib.vita_generate_actions_bf16_workspace_bytes(C.byref(fw.w), C.byref(dw.w), O, B, num_steps)
    ws = workspace(torch.device(device), nbytes)
    st = C.c_int32(0)
    L.check(lib.vita_generate_actions_bf16_status(C.byref(fw.w), O, B, num_steps, ws.data_ptr(), C.byref(st)), "vita_generate_actions_bf16_status")
    return int(st.value)


def set_sampler_spin_limit(spins: int) -> None:
    """Diagnostic: the bound of the in-launch samplers' hand-off spins (0 = the default); a tiny bound forces the time-out path."""
    L.check(L.load().vita_debug_set_spin_limit(int(spins)), "vita_debug_set_spin_limit")


def set_persistent_sampler(on: bool) -> bool:
    """Process-wide switch of vita_ode_sample_f32 / vita_generate_actions_f32 between the persistent launch and the
    per-launch path; returns the previous setting."""
    return bool(L.load().vita_ode_sample_set_persistent(1 if on else 0))


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


def generate_actions_noalloc(obs_w: Tensor, obs_b: Tensor, fw: FlowNetHandle, dw: DecoderHandle, obs_features: Tensor,
                             num_steps: int, out: Tensor, latents_out: Optional[Tensor] = None) -> Tensor:
    """vita_generate_actions_noalloc_f32: generate_actions on the library's built-in scratch array.  The caller passes every
    input and output already in the layout the C ABI wants (fp32, contiguous rows); nothing is allocated here or inside."""
    lib = L.load()
    B, O = obs_features.shape
    L.check(lib.vita_generate_actions_noalloc_f32(obs_w.data_ptr(), obs_b.data_ptr(), O, C.byref(fw.w), C.byref(dw.w),
                                                  obs_features.data_ptr(), obs_features.stride(0), out.data_ptr(), out.stride(0),
                                                  latents_out.data_ptr() if latents_out is not None else None, B, num_steps,
                                                  _stream(obs_features)),
            "vita_generate_actions_noalloc_f32")
    return out


# --------------------------------------------------------------------------------------
# alternate matchers (inference): Consistency-FM sampler, SimpleMeanFlowNet + MeanFlow sampler
# --------------------------------------------------------------------------------------
def consistency_sample(fw: FlowNetHandle, x0: Tensor, num_steps: int, eps: float = 1e-2, sigma_var: float = 1.0,
                       noise_scale: float = 1.0, noise: Optional[Tensor] = None) -> Tensor:
    """ConsistencyFlowMatcher.sample (consistency_flow_matcher.py:72-106).  `noise` (num_steps, B, D): the standard-normal
    draws the reference takes from torch.randn_like at every step (needed when sigma_var > 0)."""
    lib = L.load()
    x0 = _rows(x0, "consistency_sample: x0")
    B, D = x0.shape
    if D != fw.D:
        raise L.VitaHipError(f"consistency_sample: x0 has {D} columns, the flow net expects {fw.D}")
    if sigma_var > 0:
        if noise is None:
            raise L.VitaHipError("consistency_sample: sigma_var > 0 needs the per-step noise (num_steps, B, D)")
        noise = _f32(noise, "consistency_sample: noise").contiguous()
        if tuple(noise.shape) != (num_steps, B, D):
            raise L.VitaHipError(f"consistency_sample: noise must be {(num_steps, B, D)}, got {tuple(noise.shape)}")
    out = torch.empty((B, D), dtype=torch.float32, device=x0.device)
    nbytes = lib.vita_consistency_sample_workspace_bytes(C.byref(fw.w), B, num_steps)
    ws = workspace(x0.device, nbytes)
    L.check(lib.vita_consistency_sample_f32(C.byref(fw.w), x0.data_ptr(), _ld(x0), out.data_ptr(), D, B, num_steps, float(eps),
                                            float(sigma_var), float(noise_scale), noise.data_ptr() if noise is not None else None,
                                            ws.data_ptr(), ws.numel(), _stream(x0)), "vita_consistency_sample_f32")
    return out


class MeanFlowNetHandle:
    """vita_meanflow_weights over a state dict with SimpleMeanFlowNet's keys (simple_mean_flow_net.py:20-47)."""

    def __init__(self, sd: Mapping[str, Tensor], device, prefix: str = "flow_net.", strict: bool = False, into=None):
        device = torch.device(device)
        self.t = {}
        if strict:      # the tensors themselves (contiguous fp32 on the GPU): what a gradient descriptor needs
            g = lambda k: self.t.setdefault(k, _dev_f32(sd, prefix + k, device, True))  # noqa: E731
        else:
            g = lambda k: self.t.setdefault(k, _prep(sd, prefix + k, device))  # noqa: E731
        w = into if into is not None else L.MeanFlowWeights()
        iw = g("input_layer.weight")
        w.H, w.D = iw.shape
        n = 0
        while f"{prefix}hidden_layers.{n}.0.weight" in sd:
            n += 1
        if n > L.MAX_LAYERS:
            raise L.VitaHipError(f"MeanFlowNetHandle: {n} layers > {L.MAX_LAYERS}")
        w.L = n
        w.FF = g("hidden_layers.0.0.weight").shape[0] if n else w.H
        for a, b in (("t1", "t_embed.0"), ("t2", "t_embed.2"), ("h1", "h_embed.0"), ("h2", "h_embed.2"), ("in", "input_layer"),
                     ("u", "u_output_layer"), ("v", "v_output_layer")):
            setattr(w, a + "_w", g(b + ".weight").data_ptr())
            setattr(w, a + "_b", g(b + ".bias").data_ptr())
        for l in range(n):
            w.fc1_w[l], w.fc1_b[l] = g(f"hidden_layers.{l}.0.weight").data_ptr(), g(f"hidden_layers.{l}.0.bias").data_ptr()
            w.fc2_w[l], w.fc2_b[l] = g(f"hidden_layers.{l}.3.weight").data_ptr(), g(f"hidden_layers.{l}.3.bias").data_ptr()
        self.w, self.D, self.H, self.FF, self.L, self.device = w, w.D, w.H, w.FF, n, device


def meanflow_forward(mw: MeanFlowNetHandle, x: Tensor, t: Tensor, h: Tensor, need_v: bool = True):
    """SimpleMeanFlowNet.forward(x, timestep, h) -> (u, v) (simple_mean_flow_net.py:63-81; internal_features not returned)."""
    lib = L.load()
    x = _rows(x, "meanflow_forward: x")
    t = _f32(t, "meanflow_forward: t").contiguous()
    h = _f32(h, "meanflow_forward: h").contiguous()
    B, D = x.shape
    if D != mw.D or tuple(t.shape) != (B,) or tuple(h.shape) != (B,):
        raise L.VitaHipError(f"meanflow_forward: x {tuple(x.shape)} / t {tuple(t.shape)} / h {tuple(h.shape)} do not match D={mw.D}")
    u = torch.empty((B, D), dtype=torch.float32, device=x.device)
    v = torch.empty((B, D), dtype=torch.float32, device=x.device) if need_v else None
    nbytes = lib.vita_meanflow_forward_workspace_bytes(C.byref(mw.w), B)
    ws = workspace(x.device, nbytes)
    L.check(lib.vita_meanflow_forward_f32(C.byref(mw.w), x.data_ptr(), _ld(x), t.data_ptr(), h.data_ptr(), u.data_ptr(), D,
                                          v.data_ptr() if v is not None else None, D, B, ws.data_ptr(), ws.numel(), _stream(x)),
            "vita_meanflow_forward_f32")
    return u, v


def meanflow_sample(mw: MeanFlowNetHandle, x0: Tensor, num_steps: int = 1) -> Tensor:
    """MeanFlowMatcher.sample (mean_flow_matcher.py:201-233): x <- x - h * u(x, t, h), one C-ABI call for all steps."""
    lib = L.load()
    x0 = _rows(x0, "meanflow_sample: x0")
    B, D = x0.shape
    if D != mw.D:
        raise L.VitaHipError(f"meanflow_sample: x0 has {D} columns, the net expects {mw.D}")
    out = torch.empty((B, D), dtype=torch.float32, device=x0.device)
    nbytes = lib.vita_meanflow_sample_workspace_bytes(C.byref(mw.w), B, num_steps)
    ws = workspace(x0.device, nbytes)
    L.check(lib.vita_meanflow_sample_f32(C.byref(mw.w), x0.data_ptr(), _ld(x0), out.data_ptr(), D, B, num_steps, ws.data_ptr(),
                                         ws.numel(), _stream(x0)), "vita_meanflow_sample_f32")
    return out


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
                                  gb.data_ptr() if gb is not None else None, 0, _stream(a)), fn_name)
    return ga, gb


def mse_loss_fwd(a, b): return _loss_fwd("vita_mse_loss_fwd_f32", a, b)
def l1_loss_fwd(a, b): return _loss_fwd("vita_l1_loss_fwd_f32", a, b)
def mse_loss_bwd(a, b, g, need_a=True, need_b=True): return _loss_bwd("vita_mse_loss_bwd_f32", a, b, g, need_a, need_b)
def l1_loss_bwd(a, b, g, need_a=True, need_b=True): return _loss_bwd("vita_l1_loss_bwd_f32", a, b, g, need_a, need_b)


def contrastive_loss(a: Tensor, b: Tensor, temperature: float = 0.07, gout: Optional[Tensor] = None):
    """compute_contrastive_loss (losses/contrastive_loss.py:5-22): symmetric InfoNCE.  Returns the loss (0-d device tensor), or
    (loss, grad_a, grad_b) when `gout` (d objective / d loss) is given."""
    lib = L.load()
    a = _rows(a, "contrastive_loss: a")
    b = _rows(b, "contrastive_loss: b")
    if a.shape != b.shape:
        raise L.VitaHipError(f"contrastive_loss: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    B, D = a.shape
    out = torch.empty((), dtype=torch.float32, device=a.device)
    ga = gb = None
    if gout is not None:
        gout = _f32(gout, "contrastive_loss: gout").reshape(1).contiguous()
        ga, gb = torch.empty((B, D), dtype=torch.float32, device=a.device), torch.empty((B, D), dtype=torch.float32, device=a.device)
    nbytes = lib.vita_contrastive_loss_workspace_bytes(B, D)
    ws = workspace(a.device, nbytes)
    L.check(lib.vita_contrastive_loss_f32(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), B, D, float(temperature), out.data_ptr(),
                                          gout.data_ptr() if gout is not None else None,
                                          ga.data_ptr() if ga is not None else None, D, gb.data_ptr() if gb is not None else None, D,
                                          ws.data_ptr(), ws.numel(), _stream(a)), "vita_contrastive_loss_f32")
    return out if gout is None else (out, ga, gb)


def gaussian_kl(params: Tensor, gout: Optional[Tensor] = None):
    """DiagonalGaussianDistribution(params).kl().mean() (models/dists.py:10-31): params (B, 2C) = [mean | logvar].  With
    `gout` also returns d/d params."""
    lib = L.load()
    params = _rows(params, "gaussian_kl: params")
    B, C2 = params.shape
    if C2 % 2:
        raise L.VitaHipError(f"gaussian_kl: parameters need an even last dim (mean | logvar), got {C2}")
    Cc = C2 // 2
    ld = _ld(params)
    mean_p, logvar_p = params.data_ptr(), params.data_ptr() + 4 * Cc
    rows = torch.empty(B, dtype=torch.float32, device=params.device)
    out = torch.empty((), dtype=torch.float32, device=params.device)
    L.check(lib.vita_gaussian_kl_fwd_f32(mean_p, ld, logvar_p, ld, B, Cc, rows.data_ptr(), out.data_ptr(), _stream(params)),
            "vita_gaussian_kl_fwd_f32")
    if gout is None:
        return out
    gout = _f32(gout, "gaussian_kl: gout").reshape(1).contiguous()
    g = torch.empty((B, C2), dtype=torch.float32, device=params.device)
    L.check(lib.vita_gaussian_kl_bwd_f32(mean_p, ld, logvar_p, ld, B, Cc, gout.data_ptr(), g.data_ptr(), C2, g.data_ptr() + 4 * Cc, C2,
                                         _stream(params)), "vita_gaussian_kl_bwd_f32")
    return out, g


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


def fm_interpolate(kind: int, sigma: float, x0: Tensor, x1: Tensor, t: Tensor, eps: Optional[Tensor] = None,
                   pair_j: Optional[Tensor] = None, pair_i: Optional[Tensor] = None):
    """(xt, ut) of torchcfm's conditional / target / Schrodinger-bridge matchers (vita_fm_interpolate_f32; kind = _lib.FM_*)."""
    lib = L.load()
    x0 = _f32(x0, "fm_interpolate").contiguous()
    x1 = _f32(x1, "fm_interpolate").contiguous()
    t = _f32(t, "fm_interpolate").contiguous()
    if eps is not None:
        eps = _f32(eps, "fm_interpolate: eps").contiguous()
        if eps.shape != x0.shape:
            raise L.VitaHipError(f"fm_interpolate: eps {tuple(eps.shape)} must match x0 {tuple(x0.shape)}")
    B, D = x0.shape
    pair_i, pair_j = _idx(pair_i, x0.device), _idx(pair_j, x0.device)
    xt = torch.empty_like(x0)
    ut = torch.empty_like(x0)
    L.check(lib.vita_fm_interpolate_f32(int(kind), float(sigma), x0.data_ptr(), x1.data_ptr(), t.data_ptr(),
                                        eps.data_ptr() if eps is not None else None,
                                        pair_i.data_ptr() if pair_i is not None else None,
                                        pair_j.data_ptr() if pair_j is not None else None,
                                        xt.data_ptr(), ut.data_ptr(), B, D, _stream(x0)), "vita_fm_interpolate_f32")
    return xt, ut


def fm_interpolate_bwd(kind: int, sigma: float, dxt: Tensor, dut: Tensor, t: Tensor, pair_j=None, pair_i=None):
    lib = L.load()
    dxt, dut, t = dxt.contiguous(), dut.contiguous(), t.contiguous()
    B, D = dxt.shape
    pair_i, pair_j = _idx(pair_i, dxt.device), _idx(pair_j, dxt.device)
    d0, d1 = torch.empty_like(dxt), torch.empty_like(dxt)
    L.check(lib.vita_fm_interpolate_bwd_f32(int(kind), float(sigma), dxt.data_ptr(), dut.data_ptr(), t.data_ptr(),
                                            pair_i.data_ptr() if pair_i is not None else None,
                                            pair_j.data_ptr() if pair_j is not None else None,
                                            d0.data_ptr(), d1.data_ptr(), B, D, 0, _stream(dxt)), "vita_fm_interpolate_bwd_f32")
    return d0, d1


def _norm(fn_name: str, x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Stats of shape (C) act on the last dim; stats of shape (C,1,..,1) act on dim -mean.dim() (image statistics)."""
    lib = L.load()
    x = _f32(x, fn_name).contiguous()
    trailing = 0
    while mean.dim() > 1 + trailing and mean.shape[-1 - trailing] == 1:
        trailing += 1
    if mean.dim() != 1 + trailing or std.shape != mean.shape:
        raise L.VitaHipError(f"{fn_name}: statistics must be (C) or (C,1,..,1), got {tuple(mean.shape)} / {tuple(std.shape)}")
    Cdim = mean.shape[0]
    if x.dim() < 1 + trailing or x.shape[x.dim() - 1 - trailing] != Cdim:
        raise L.VitaHipError(f"{fn_name}: x {tuple(x.shape)} does not match statistics {tuple(mean.shape)}")
    inner = 1
    for d in x.shape[x.dim() - trailing:]:
        inner *= d
    mean = _f32(mean, fn_name).contiguous().reshape(-1)
    std = _f32(std, fn_name).contiguous().reshape(-1)
    y = torch.empty_like(x)
    L.check(getattr(lib, fn_name)(x.data_ptr(), mean.data_ptr(), std.data_ptr(), y.data_ptr(), x.numel() // (Cdim * inner), Cdim,
                                  inner, _stream(x)), fn_name)
    return y


class _ChannelAffineReLU(torch.autograd.Function):
    """vita_channel_affine_relu_f32 / _bwd_f32 (include/vita_hip.h)."""

    @staticmethod
    def forward(ctx, x, res, scale, shift, relu):
        lib = L.load()
        x = _f32(x, "channel_affine_relu: x").contiguous()
        if x.dim() < 2 or scale.shape != (x.shape[1],) or shift.shape != scale.shape:
            raise L.VitaHipError(f"channel_affine_relu: x {tuple(x.shape)} against scale {tuple(scale.shape)} / shift {tuple(shift.shape)}")
        if res is not None:
            res = _f32(res, "channel_affine_relu: res").contiguous()
            if res.shape != x.shape:
                raise L.VitaHipError(f"channel_affine_relu: shortcut {tuple(res.shape)} against x {tuple(x.shape)}")
        scale = _f32(scale, "channel_affine_relu: scale").contiguous()
        shift = _f32(shift, "channel_affine_relu: shift").contiguous()
        for name, t in (("scale", scale), ("shift", shift), ("shortcut", res)):
            if t is not None and t.device != x.device:
                raise L.VitaHipError(f"channel_affine_relu: {name} is on {t.device}, x on {x.device}")
        Cn = x.shape[1]
        hw = 1
        for d in x.shape[2:]:
            hw *= d
        y = torch.empty_like(x)
        L.check(lib.vita_channel_affine_relu_f32(x.data_ptr(), res.data_ptr() if res is not None else None, scale.data_ptr(), shift.data_ptr(),
                                                 y.data_ptr(), x.shape[0] * Cn, Cn, hw, int(relu), _stream(x)), "vita_channel_affine_relu_f32")
        ctx.relu, ctx.has_res, ctx.dims = bool(relu), res is not None, (x.shape[0] * Cn, Cn, hw)
        ctx.save_for_backward(y if relu else None, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        y, scale = ctx.saved_tensors
        dy = _f32(dy, "channel_affine_relu: dy").contiguous()
        dx = torch.empty_like(dy)
        dres = torch.empty_like(dy) if ctx.has_res and ctx.needs_input_grad[1] else None
        rows, Cn, hw = ctx.dims
        L.check(lib.vita_channel_affine_relu_bwd_f32(dy.data_ptr(), y.data_ptr() if y is not None else None, scale.data_ptr(), dx.data_ptr(),
                                                     dres.data_ptr() if dres is not None else None, rows, Cn, hw, int(ctx.relu), _stream(dy)),
                "vita_channel_affine_relu_bwd_f32")
        return dx, dres, None, None, None


def channel_affine_relu(x: Tensor, scale: Tensor, shift: Tensor, res: Optional[Tensor] = None, relu: bool = True) -> Tensor:
    """act((x * scale[c] + shift[c]) + res) over (N, C, ...) in one pass, with its backward (frozen batch norm + shortcut + ReLU of the
    observer's ResNet18, vita_amd/observers.py)."""
    return _ChannelAffineReLU.apply(x, res, scale, shift, relu)


class _AffineReLUMaxPool(torch.autograd.Function):
    """vita_affine_relu_maxpool3s2_f32 / _bwd_f32 (include/vita_hip.h)."""
    MAX_PLANES = 65535

    @staticmethod
    def forward(ctx, x, scale, shift):
        lib = L.load()
        x = _f32(x, "affine_relu_maxpool: x").contiguous()
        if x.dim() != 4 or scale.shape != (x.shape[1],) or shift.shape != scale.shape:
            raise L.VitaHipError(f"affine_relu_maxpool: x {tuple(x.shape)} against scale {tuple(scale.shape)} / shift {tuple(shift.shape)}")
        scale = _f32(scale, "affine_relu_maxpool: scale").contiguous()
        shift = _f32(shift, "affine_relu_maxpool: shift").contiguous()
        if scale.device != x.device or shift.device != x.device:
            raise L.VitaHipError(f"affine_relu_maxpool: scale / shift on {scale.device} / {shift.device}, x on {x.device}")
        N, Cn, H, W = x.shape
        if H < 1 or W < 1:
            raise L.VitaHipError(f"affine_relu_maxpool: empty image {tuple(x.shape)}")
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, Cn, Ho, Wo), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, Cn, Ho, Wo), dtype=torch.uint8, device=x.device)
        per = max(1, _AffineReLUMaxPool.MAX_PLANES // Cn)          # images per launch (grid z)
        for n0 in range(0, N, per):
            n1 = min(N, n0 + per)
            L.check(lib.vita_affine_relu_maxpool3s2_f32(x[n0:n1].data_ptr(), scale.data_ptr(), shift.data_ptr(), y[n0:n1].data_ptr(),
                                                        idx[n0:n1].data_ptr(), (n1 - n0) * Cn, Cn, H, W, _stream(x)),
                    "vita_affine_relu_maxpool3s2_f32")
        ctx.dims = (N, Cn, H, W, per)
        ctx.save_for_backward(idx, scale)
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        lib = L.load()
        idx, scale = ctx.saved_tensors
        N, Cn, H, W, per = ctx.dims
        dy = _f32(dy, "affine_relu_maxpool: dy").contiguous()
        dx = torch.empty((N, Cn, H, W), dtype=torch.float32, device=dy.device)
        for n0 in range(0, N, per):
            n1 = min(N, n0 + per)
            L.check(lib.vita_affine_relu_maxpool3s2_bwd_f32(dy[n0:n1].data_ptr(), idx[n0:n1].data_ptr(), scale.data_ptr(), dx[n0:n1].data_ptr(),
                                                            (n1 - n0) * Cn, Cn, H, W, _stream(dy)),
                    "vita_affine_relu_maxpool3s2_bwd_f32")
        return dx, None, None


def affine_relu_maxpool(x: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    """max_pool2d(relu(x * scale[c] + shift[c]), 3, 2, 1) over (N, C, H, W) in one pass, with its backward (the stem of the observer's
    ResNet18, vita_amd/observers.py)."""
    return _AffineReLUMaxPool.apply(x, scale, shift)[0]


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


# --------------------------------------------------------------------------------------
# training side: backward kernels (thin wrappers, used by the tests and the trainer)
# --------------------------------------------------------------------------------------
def linear_t(x: Tensor, w: Tensor, *, a_col: bool = False, b_col: bool = False, M: int, N: int, K: int,
             bias: Optional[Tensor] = None, gate_scalar: Optional[float] = None, resid: Optional[Tensor] = None,
             act: int = L.ACT_NONE, act_grad_z: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """vita_linear_f32 with transposed operands: C[M,N] = resid + gs * act'(z) * (sum_k A(m,k) B(n,k) + bias) where
    A(m,k) = x[k,m] if a_col else x[m,k], B(n,k) = w[k,n] if b_col else w[n,k] (dgrad / wgrad of nn.Linear)."""
    lib = L.load()
    x = _rows(x, "linear_t: x")
    w = _rows(w, "linear_t: w")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    a = L.LinearArgs()
    a.X, a.ldx, a.W, a.ldw = x.data_ptr(), _ld(x), w.data_ptr(), _ld(w)
    a.bias = bias.data_ptr() if bias is not None else None
    a.Y, a.ldy, a.M, a.N, a.K = out.data_ptr(), _ld(out), M, N, K
    a.a_col, a.b_col, a.act = int(a_col), int(b_col), act
    if act_grad_z is not None:
        z = _rows(act_grad_z, "linear_t: act_grad_z")
        a.act_grad_z, a.ld_act_grad_z = z.data_ptr(), _ld(z)
    if gate_scalar is not None:
        a.use_gate_scalar, a.gate_scalar = 1, float(gate_scalar)
    if resid is not None:
        a.resid, a.ld_resid = resid.data_ptr(), _ld(resid)
    _splitk_ws(x.device, a)
    L.check(lib.vita_linear_f32(C.byref(a), _stream(x)), "vita_linear_f32 (transposed)")
    return out


def act_bwd(z: Tensor, dy: Tensor, act: int) -> Tensor:
    lib = L.load()
    z, dy = _f32(z, "act_bwd").contiguous(), _f32(dy, "act_bwd").contiguous()
    dz = torch.empty_like(z)
    L.check(lib.vita_act_bwd_f32(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), z.numel(), act, _stream(z)), "vita_act_bwd_f32")
    return dz


def act_fwd(z: Tensor, act: int) -> Tensor:
    lib = L.load()
    z = _f32(z, "act_fwd").contiguous()
    y = torch.empty_like(z)
    L.check(lib.vita_act_fwd_f32(z.data_ptr(), y.data_ptr(), z.numel(), act, _stream(z)), "vita_act_fwd_f32")
    return y


def gate_bwd(dy: Tensor, z: Tensor, gate, need_dgate: bool = True):
    """y = r + z*g: returns (dz, dgate_rows).  gate: (N) / (M,N) tensor or a python float."""
    lib = L.load()
    dy, z = _f32(dy, "gate_bwd").contiguous(), _f32(z, "gate_bwd").contiguous()
    M, N = dy.shape
    dz = torch.empty_like(dy)
    dg = torch.empty_like(dy) if need_dgate else None
    if isinstance(gate, float):
        gp, ldg, gs = None, 0, gate
    else:
        gate = gate.contiguous()
        gp, ldg, gs = gate.data_ptr(), (0 if gate.dim() == 1 else N), 0.0
    L.check(lib.vita_gate_bwd_f32(dy.data_ptr(), z.data_ptr(), gp, ldg, gs, dz.data_ptr(),
                                  dg.data_ptr() if dg is not None else None, N, M, N, _stream(dy)), "vita_gate_bwd_f32")
    return dz, dg


def layernorm_modulate_bwd(x: Tensor, dy: Tensor, mul: Tensor, eps: float, mode: int, dx_init: Optional[Tensor] = None):
    """Backward of layernorm_modulate: returns (dx [+ dx_init], dmul_rows, dadd_rows)."""
    lib = L.load()
    x, dy = _rows(x, "ln_bwd: x"), _rows(dy, "ln_bwd: dy")
    M, H = x.shape
    mul = mul.contiguous()
    dx = dx_init.clone() if dx_init is not None else torch.empty_like(x)
    dm, da = torch.empty((M, H), device=x.device), torch.empty((M, H), device=x.device)
    L.check(lib.vita_layernorm_modulate_bwd_f32(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), mul.data_ptr(),
                                                0 if mul.dim() == 1 else H, M, H, eps, mode, dx.data_ptr(), H,
                                                1 if dx_init is not None else 0, dm.data_ptr(), da.data_ptr(), H, _stream(x)),
            "vita_layernorm_modulate_bwd_f32")
    return dx, dm, da


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    lib = L.load()
    x = _rows(x, "colsum: x")
    M, N = x.shape
    if out is None:
        if accumulate:
            raise L.VitaHipError("colsum: accumulate=True needs `out`")
        out = torch.empty(N, dtype=torch.float32, device=x.device)     # (written, not accumulated: no fill launch in front)
    part = torch.empty(lib.vita_colsum_partials_count(N), dtype=torch.float32, device=x.device)
    L.check(lib.vita_colsum_f32(x.data_ptr(), _ld(x), M, N, part.data_ptr(), out.data_ptr(), int(accumulate), _stream(x)),
            "vita_colsum_f32")
    return out


def im2col_k5s2(x: Tensor) -> Tensor:
    lib = L.load()
    x = _f32(x, "im2col").contiguous()
    B, Lin, Cc = x.shape
    Lout = (Lin - 1) // 2 + 1
    col = torch.empty((B * Lout, Cc * 5), dtype=torch.float32, device=x.device)
    L.check(lib.vita_im2col_k5s2_f32(x.data_ptr(), col.data_ptr(), B, Lin, Cc, _stream(x)), "vita_im2col_k5s2_f32")
    return col


def col2im_k5s2(dcol: Tensor, B: int, Lin: int, Cc: int) -> Tensor:
    lib = L.load()
    dcol = _f32(dcol, "col2im").contiguous()
    dx = torch.empty((B, Lin, Cc), dtype=torch.float32, device=dcol.device)
    L.check(lib.vita_col2im_k5s2_f32(dcol.data_ptr(), dx.data_ptr(), B, Lin, Cc, _stream(dcol)), "vita_col2im_k5s2_f32")
    return dx


def nlc_ncl(x: Tensor, to_ncl: bool) -> Tensor:
    lib = L.load()
    x = _f32(x, "nlc_ncl").contiguous()
    if to_ncl:
        B, Ln, Cc = x.shape
        out = torch.empty((B, Cc, Ln), dtype=torch.float32, device=x.device)
    else:
        B, Cc, Ln = x.shape
        out = torch.empty((B, Ln, Cc), dtype=torch.float32, device=x.device)
    L.check(lib.vita_nlc_ncl_f32(x.data_ptr(), out.data_ptr(), B, Ln, Cc, int(to_ncl), _stream(x)), "vita_nlc_ncl_f32")
    return out


def cfm_interpolate_bwd(dxt: Tensor, dut: Tensor, t: Tensor, pair_j=None, pair_i=None):
    lib = L.load()
    dxt, dut, t = dxt.contiguous(), dut.contiguous(), t.contiguous()
    B, D = dxt.shape
    pair_i, pair_j = _idx(pair_i, dxt.device), _idx(pair_j, dxt.device)
    d0, d1 = torch.empty_like(dxt), torch.empty_like(dxt)
    L.check(lib.vita_cfm_interpolate_bwd_f32(dxt.data_ptr(), dut.data_ptr(), t.data_ptr(),
                                             pair_i.data_ptr() if pair_i is not None else None,
                                             pair_j.data_ptr() if pair_j is not None else None,
                                             d0.data_ptr(), d1.data_ptr(), B, D, 0, _stream(dxt)), "vita_cfm_interpolate_bwd_f32")
    return d0, d1


def sumsq(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    lib = L.load()
    x = _f32(x, "sumsq").contiguous()
    part = torch.empty(lib.vita_loss_partials_count(), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=x.device)
    L.check(lib.vita_sumsq_f32(x.data_ptr(), x.numel(), part.data_ptr(), out.data_ptr(), _stream(x)), "vita_sumsq_f32")
    return out


def adamw_ema_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, ema: Optional[Tensor], *, step: int, lr: float,
                   betas=(0.95, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6, gnorm_sq: Optional[Tensor] = None,
                   max_norm: float = 0.0, grad_scale: float = 1.0, ema_decay: float = 0.0) -> None:
    """In-place fused clip + AdamW + EMA over flat fp32 buffers (see vita_adamw_ema_step_f32)."""
    lib = L.load()
    for t_ in (p, g, m, v) + ((ema,) if ema is not None else ()):
        _f32(t_, "adamw_ema_step")
        if not t_.is_contiguous():
            raise L.VitaHipError("adamw_ema_step: buffers must be contiguous")
    L.check(lib.vita_adamw_ema_step_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                        ema.data_ptr() if ema is not None else None, p.numel(),
                                        gnorm_sq.data_ptr() if gnorm_sq is not None else None, max_norm, grad_scale, lr,
                                        betas[0], betas[1], eps, weight_decay, step, ema_decay, _stream(p)),
            "vita_adamw_ema_step_f32")


# --------------------------------------------------------------------------------------
# whole-policy descriptor and the fused training step
# --------------------------------------------------------------------------------------
def _dev_f32(sd: Mapping[str, Tensor], key: str, device, strict: bool) -> Tensor:
    t = sd[key]
    if strict:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise L.VitaHipError(f"{key}: gradient / training tensors must already be contiguous fp32 on the GPU")
        return t.detach()
    return _prep(sd, key, device)


def _fill_flow(w, g, prefix: str, sd) -> None:
    in_w = g(prefix + "input_proj.weight")
    w.H, w.D = in_w.shape
    w.TE = g(prefix + "time_embed.1.weight").shape[1]
    n = 0
    while f"{prefix}layers.{n}.mlp.fc1.weight" in sd:
        n += 1
    if n > L.MAX_LAYERS:
        raise L.VitaHipError(f"flow net has {n} layers; the ABI supports {L.MAX_LAYERS}")
    w.L = n
    w.FF = g(prefix + "layers.0.mlp.fc1.weight").shape[0] if n else w.H
    w.in_w, w.in_b = in_w.data_ptr(), g(prefix + "input_proj.bias").data_ptr()
    w.te1_w, w.te1_b = g(prefix + "time_embed.1.weight").data_ptr(), g(prefix + "time_embed.1.bias").data_ptr()
    w.te3_w, w.te3_b = g(prefix + "time_embed.3.weight").data_ptr(), g(prefix + "time_embed.3.bias").data_ptr()
    for l in range(n):
        w.fc1_w[l] = g(f"{prefix}layers.{l}.mlp.fc1.weight").data_ptr()
        w.fc1_b[l] = g(f"{prefix}layers.{l}.mlp.fc1.bias").data_ptr()
        w.fc2_w[l] = g(f"{prefix}layers.{l}.mlp.fc2.weight").data_ptr()
        w.fc2_b[l] = g(f"{prefix}layers.{l}.mlp.fc2.bias").data_ptr()
        w.mod_w[l] = g(f"{prefix}layers.{l}.time_modulator.1.weight").data_ptr()
        w.mod_b[l] = g(f"{prefix}layers.{l}.time_modulator.1.bias").data_ptr()
    w.norm_w, w.norm_b = g(prefix + "norm.weight").data_ptr(), g(prefix + "norm.bias").data_ptr()
    w.out_w, w.out_b = g(prefix + "out_proj.weight").data_ptr(), g(prefix + "out_proj.bias").data_ptr()


def _fill_dec(w, g, prefix: str, sd) -> None:
    w.HD, w.D = g(prefix + "input_proj.weight").shape
    w.OUT = g(prefix + "output_proj.weight").shape[0]
    n = 0
    while f"{prefix}layers.{n}.fc1.weight" in sd:
        n += 1
    w.NL = n
    w.in_w, w.in_b = g(prefix + "input_proj.weight").data_ptr(), g(prefix + "input_proj.bias").data_ptr()
    for i in range(n):
        w.fc1_w[i] = g(f"{prefix}layers.{i}.fc1.weight").data_ptr()
        w.fc1_b[i] = g(f"{prefix}layers.{i}.fc1.bias").data_ptr()
        w.fc2_w[i] = g(f"{prefix}layers.{i}.fc2.weight").data_ptr()
        w.fc2_b[i] = g(f"{prefix}layers.{i}.fc2.bias").data_ptr()
    w.out_w, w.out_b = g(prefix + "output_proj.weight").data_ptr(), g(prefix + "output_proj.bias").data_ptr()


def _fill_enc(w, g, prefix: str, sd, pred_horizon: int, latent_dim: int) -> None:
    w.T, w.D = pred_horizon, latent_dim
    if prefix + "latent_proj.weight" in sd:        # CNNActionEncoder, action_ae.py:282-327
        w.kind = L.ENC_CNN
        n = 0
        while f"{prefix}encoder.{2 * n}.weight" in sd:
            n += 1
        w.NL = n
        c0 = g(f"{prefix}encoder.0.weight")
        w.C, w.A = c0.shape[0], c0.shape[1]
        for i in range(n):
            w.w1[i] = g(f"{prefix}encoder.{2 * i}.weight").data_ptr()
            w.b1[i] = g(f"{prefix}encoder.{2 * i}.bias").data_ptr()
        pw_ = g(prefix + "latent_proj.weight")
        w.proj_w, w.proj_b = pw_.data_ptr(), g(prefix + "latent_proj.bias").data_ptr()
        if pw_.shape[0] == 2 * latent_dim:         # variational CNNActionEncoder, action_vae.py:213
            w.variational = 1
        elif pw_.shape[0] != latent_dim:
            raise L.VitaHipError(f"action_encoder.latent_proj has {pw_.shape[0]} outputs; expected latent_dim={latent_dim} or twice that")
    else:                                          # SimpleActionEncoder, :111-170
        w.kind = L.ENC_SIMPLE
        pw = g(prefix + "input_proj.weight")
        w.C, w.A = pw.shape
        n = 0
        while f"{prefix}layers.{n}.fc1.weight" in sd:
            n += 1
        w.NL = n
        for i in range(n):
            w.w1[i] = g(f"{prefix}layers.{i}.fc1.weight").data_ptr()
            w.b1[i] = g(f"{prefix}layers.{i}.fc1.bias").data_ptr()
            w.w2[i] = g(f"{prefix}layers.{i}.fc2.weight").data_ptr()
            w.b2[i] = g(f"{prefix}layers.{i}.fc2.bias").data_ptr()
        w.proj_w, w.proj_b = pw.data_ptr(), g(prefix + "input_proj.bias").data_ptr()


class PolicyHandle:
    """vita_policy_weights over a state dict with the reference's keys.  With strict=True the
    tensors are used in place (they must be contiguous fp32 on the GPU): that is what a gradient descriptor needs,
    since the kernels write into them."""

    def __init__(self, sd: Mapping[str, Tensor], device, pred_horizon: int = 16, strict: bool = False,
                 external_encoder_action_dim: Optional[int] = None, external_decoder_action_dim: Optional[int] = None):
        """external_encoder_action_dim: the action encoder is the caller's own module (VITA_ENC_EXTERNAL): `sd` holds no
        action_encoder.* tensors, compute_loss takes the encoder's output as `enc_latents`.
        external_decoder_action_dim: the action decoder is the caller's own module (vita_loss_config.external_decoder): `sd`
        holds no action_decoder.* tensors; compute_loss then runs as a forward call that returns [pred; z1] and a backward call
        that takes d (decoder terms) / d [pred; z1]."""
        device = torch.device(device)
        self.device = device
        self.t = {}
        g = lambda k: self.t.setdefault(k, _dev_f32(sd, k, device, strict))  # noqa: E731
        w = L.PolicyWeights()
        ow = g("obs_encoder.weight")
        w.O = ow.shape[1]
        w.obs_w, w.obs_b = ow.data_ptr(), g("obs_encoder.bias").data_ptr()
        _fill_flow(w.flow, g, "flow_net.", sd)
        self.external_decoder = external_decoder_action_dim is not None
        self._step_ws = None        # external decoder: the workspace that lives from the forward call to the backward call
        if self.external_decoder:
            w.dec.D, w.dec.OUT = ow.shape[0], pred_horizon * int(external_decoder_action_dim)
        else:
            _fill_dec(w.dec, g, "action_decoder.", sd)
        if external_encoder_action_dim is not None:
            w.enc.kind, w.enc.A, w.enc.T, w.enc.D = L.ENC_EXTERNAL, int(external_encoder_action_dim), pred_horizon, ow.shape[0]
        else:
            _fill_enc(w.enc, g, "action_encoder.", sd, pred_horizon, ow.shape[0])
        self.w = w
        self.O, self.D, self.T, self.A, self.OUT = w.O, w.flow.D, w.enc.T, w.enc.A, w.dec.OUT


def loss_config(num_steps: int = 6, recon: str = "l1", decode_flow_latents: bool = True, consistency_weight: float = 1.0,
                enc_recon_weight: float = 0.5, flow_recon_weight: float = 0.5, enc_contrastive_weight: float = 0.0,
                flow_contrastive_weight: float = 0.0, temperature: float = 0.07, freeze_encoder: bool = False,
                freeze_decoder: bool = False, kl_weight: float = 0.0, gemm_bf16: int = 0, fm_kind: int = 0,
                fm_sigma: float = 0.0, side_streams: bool = False) -> "L.LossConfig":
    if recon not in ("l1", "l2"):
        raise ValueError(f"Unsupported recon_loss_type: {recon}. Use 'l1' or 'l2'.")   # vita_policy.py:71
    c = L.LossConfig()
    c.num_steps, c.recon_l1, c.decode_flow_latents = num_steps, int(recon == "l1"), int(decode_flow_latents)
    c.consistency_weight, c.enc_recon_weight, c.flow_recon_weight = consistency_weight, enc_recon_weight, flow_recon_weight
    c.enc_contrastive_weight, c.flow_contrastive_weight, c.temperature = enc_contrastive_weight, flow_contrastive_weight, temperature
    c.freeze_encoder, c.freeze_decoder = int(freeze_encoder), int(freeze_decoder)
    c.kl_weight, c.gemm_bf16 = float(kl_weight), int(gemm_bf16)
    c.fm_kind, c.fm_sigma = int(fm_kind), float(fm_sigma)
    c.side_streams = int(side_streams)
    return c


def _event_handles(events, who: str):
    """torch.cuda.Event list -> C array of hipEvent_t (None entries stay NULL), or None."""
    if events is None:
        return None
    handles = []
    for e in events:
        h = e.cuda_event if e is not None else None
        if e is not None and not h:
            raise L.VitaHipError(f"{who}: grad_ready event has no hipEvent_t yet (torch creates it on the first "
                                 "record(): call event.record() once before passing it)")
        handles.append(h)
    return (L.c_fp * len(handles))(*handles)


def compute_loss(pw: PolicyHandle, cfg, obs_features: Tensor, actions: Tensor, t: Tensor, *, grads: Optional[PolicyHandle] = None,
                 pair_i=None, pair_j=None, need_dobs: bool = False, grad_ready_events=None, losses: Optional[Tensor] = None,
                 enc_noise: Optional[Tensor] = None, fm_noise: Optional[Tensor] = None, enc_latents: Optional[Tensor] = None,
                 dec_dlatents: Optional[Tensor] = None):
    """VitaPolicy.compute_loss (+ backward when `grads` is given), vita_policy.py:112-189.  Returns
    (losses (8,) device tensor indexed as _lib.LOSS_NAMES, d_obs_features or None).  No host sync.
    enc_latents: with an external action encoder (PolicyHandle(external_encoder_action_dim=...)) the (B, D) output of the
    caller's module; the return value then has a third entry, d loss / d enc_latents (None without `grads`).
    With an external action decoder (PolicyHandle(external_decoder_action_dim=...)) the step is two calls with the same
    arguments: without `grads` the forward, whose return value ends with the (2B, D) latents [pred; z1] the caller decodes (the
    reconstruction terms are the caller's; losses[TOTAL] is the sum of the others); then with `grads`, `losses` (left as the
    forward wrote it) and dec_dlatents = d (caller's terms) / d latents, the backward alone."""
    lib = L.load()
    obs_features = _rows(obs_features, "compute_loss: obs_features")
    actions = _f32(actions, "compute_loss: actions").contiguous()
    t = _f32(t, "compute_loss: t").contiguous()
    B = obs_features.shape[0]
    if obs_features.shape[1] != pw.O or tuple(actions.shape) != (B, pw.T, pw.A) or tuple(t.shape) != (B,):
        raise L.VitaHipError(f"compute_loss: obs {tuple(obs_features.shape)} / actions {tuple(actions.shape)} / t {tuple(t.shape)} "
                             f"do not match O={pw.O}, T={pw.T}, A={pw.A}")
    dev = obs_features.device
    pair_i, pair_j = _idx(pair_i, dev), _idx(pair_j, dev)
    if enc_noise is not None:
        enc_noise = _f32(enc_noise, "compute_loss: enc_noise").contiguous()
        if tuple(enc_noise.shape) != (B, pw.D):
            raise L.VitaHipError(f"compute_loss: enc_noise {tuple(enc_noise.shape)} must be (B, D) = ({B}, {pw.D})")
    if fm_noise is not None:
        fm_noise = _f32(fm_noise, "compute_loss: fm_noise").contiguous()
        if tuple(fm_noise.shape) != (B, pw.D):
            raise L.VitaHipError(f"compute_loss: fm_noise {tuple(fm_noise.shape)} must be (B, D) = ({B}, {pw.D})")
    elif cfg.fm_kind != L.FM_CONDITIONAL or cfg.fm_sigma != 0.0:
        raise L.VitaHipError("compute_loss: this flow matcher adds noise to xt: pass fm_noise (B, D)")
    if losses is None:
        losses = torch.empty(L.LOSS_COUNT, dtype=torch.float32, device=dev)
    elif losses.numel() < L.LOSS_COUNT:
        raise L.VitaHipError(f"compute_loss: losses needs {L.LOSS_COUNT} floats")
    dobs = torch.empty((B, pw.O), dtype=torch.float32, device=dev) if (need_dobs and grads is not None) else None
    ev = _event_handles(grad_ready_events, "compute_loss")
    if ev is not None and len(ev) > 3:      # 4 + L events: per-section readiness of the flow net's gradients (vita_loss_config.grad_ready_layers)
        if len(ev) != 4 + pw.w.flow.L:
            raise L.VitaHipError(f"compute_loss: {len(ev)} grad_ready events; pass 3, or 4 + L = {4 + pw.w.flow.L}")
        cfg = L.LossConfig.from_buffer_copy(cfg)
        cfg.grad_ready_layers = 1
    external = pw.w.enc.kind == L.ENC_EXTERNAL
    dlat = None
    wdesc, gdesc = pw.w, (grads.w if grads is not None else None)
    if external != (enc_latents is not None):
        raise L.VitaHipError("compute_loss: enc_latents goes with an external action encoder, and only with one")
    if external:
        enc_latents = _f32(enc_latents, "compute_loss: enc_latents").contiguous()
        if tuple(enc_latents.shape) != (B, pw.D):
            raise L.VitaHipError(f"compute_loss: enc_latents {tuple(enc_latents.shape)} must be (B, D) = ({B}, {pw.D})")
        wdesc = L.PolicyWeights.from_buffer_copy(pw.w)
        wdesc.enc.proj_w = enc_latents.data_ptr()
        if grads is not None:
            dlat = torch.empty((B, pw.D), dtype=torch.float32, device=dev)
            gdesc = L.PolicyWeights.from_buffer_copy(grads.w)
            gdesc.enc.proj_w = dlat.data_ptr()
    latents = None
    if pw.external_decoder:
        if grads is not None and not grads.external_decoder:
            raise L.VitaHipError("compute_loss: the gradient handle must be built with external_decoder_action_dim too")
        cfg = L.LossConfig.from_buffer_copy(cfg)
        cfg.external_decoder = 1
        if wdesc is pw.w:
            wdesc = L.PolicyWeights.from_buffer_copy(pw.w)
        if grads is None:
            if dec_dlatents is not None:
                raise L.VitaHipError("compute_loss: dec_dlatents goes with the backward call (grads)")
            latents = torch.empty((2 * B, pw.D), dtype=torch.float32, device=dev)
            wdesc.dec.in_w = latents.data_ptr()
        else:
            if dec_dlatents is None or tuple(dec_dlatents.shape) != (2 * B, pw.D):
                raise L.VitaHipError(f"compute_loss: the backward call of an external decoder needs dec_dlatents (2B, D) = ({2 * B}, {pw.D})")
            dec_dlatents = _f32(dec_dlatents, "compute_loss: dec_dlatents").contiguous()
            cfg.resume_backward = 1
            wdesc.dec.in_w = dec_dlatents.data_ptr()     # not written by a backward call; the descriptor check wants it non-null
            if gdesc is grads.w:
                gdesc = L.PolicyWeights.from_buffer_copy(grads.w)
            gdesc.dec.in_w = dec_dlatents.data_ptr()
    elif dec_dlatents is not None:
        raise L.VitaHipError("compute_loss: dec_dlatents goes with an external action decoder, and only with one")
    nbytes = lib.vita_compute_loss_workspace_bytes(C.byref(wdesc), C.byref(cfg), B)
    if pw.external_decoder:     # the saved activations must survive until the backward call: a workspace of the handle's own
        if grads is None:
            if pw._step_ws is None or pw._step_ws.numel() < nbytes or pw._step_ws.device != dev:
                pw._step_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        elif pw._step_ws is None or pw._step_ws.numel() < nbytes:
            raise L.VitaHipError("compute_loss: the backward call of an external decoder must follow its forward call (same batch size)")
        ws = pw._step_ws
    else:
        ws = workspace(dev, nbytes)
    L.check(lib.vita_compute_loss_f32(C.byref(wdesc), C.byref(gdesc) if gdesc is not None else None, C.byref(cfg),
                                      obs_features.data_ptr(), _ld(obs_features), actions.data_ptr(), t.data_ptr(),
                                      pair_i.data_ptr() if pair_i is not None else None,
                                      pair_j.data_ptr() if pair_j is not None else None,
                                      fm_noise.data_ptr() if fm_noise is not None else None,
                                      enc_noise.data_ptr() if enc_noise is not None else None, B, losses.data_ptr(),
                                      dobs.data_ptr() if dobs is not None else None, pw.O, ev, ws.data_ptr(), ws.numel(),
                                      _stream(obs_features)), "vita_compute_loss_f32")
    out = (losses, dobs, dlat) if external else (losses, dobs)
    if pw.external_decoder and grads is None:
        out = out + (latents,)
    return out


class MeanFlowPolicyHandle:
    """vita_meanflow_policy_weights over a state dict whose flow_net.* keys are SimpleMeanFlowNet's (policy.flow_net.name:
    simple_mean_flow_net); the other modules as in PolicyHandle."""

    def __init__(self, sd: Mapping[str, Tensor], device, pred_horizon: int = 16, strict: bool = False):
        device = torch.device(device)
        self.device = device
        self.t = {}
        g = lambda k: self.t.setdefault(k, _dev_f32(sd, k, device, strict))  # noqa: E731
        w = L.MeanFlowPolicyWeights()
        ow = g("obs_encoder.weight")
        w.O = ow.shape[1]
        w.obs_w, w.obs_b = ow.data_ptr(), g("obs_encoder.bias").data_ptr()
        self.flow = MeanFlowNetHandle(sd, device, "flow_net.", strict=strict, into=w.flow)
        _fill_dec(w.dec, g, "action_decoder.", sd)
        _fill_enc(w.enc, g, "action_encoder.", sd, pred_horizon, ow.shape[0])
        self.w = w
        self.O, self.D, self.T, self.A, self.OUT = w.O, w.flow.D, w.enc.T, w.enc.A, w.dec.OUT


def meanflow_loss_config(num_steps: int = 1, recon: str = "l1", decode_flow_latents: bool = True, consistency_weight: float = 1.0,
                         enc_recon_weight: float = 0.5, flow_recon_weight: float = 0.5, adaptive_loss_gamma: float = 0.5,
                         adaptive_loss_c: float = 1e-3, use_imf: bool = False, norm_p: float = 1.0, norm_eps: float = 0.01,
                         aux_v_loss_weight: float = 1.0, dispersive_loss_weight: float = 0.0,
                         dispersive_loss_tau: float = 1.0, freeze_encoder: bool = False,
                         freeze_decoder: bool = False, enc_contrastive_weight: float = 0.0, flow_contrastive_weight: float = 0.0,
                         temperature: float = 0.07, kl_weight: float = 0.0) -> "L.MeanFlowLossConfig":
    if recon not in ("l1", "l2"):
        raise ValueError(f"Unsupported recon_loss_type: {recon}. Use 'l1' or 'l2'.")
    c = L.MeanFlowLossConfig()
    c.num_steps, c.recon_l1, c.decode_flow_latents = num_steps, int(recon == "l1"), int(decode_flow_latents)
    c.consistency_weight, c.enc_recon_weight, c.flow_recon_weight = consistency_weight, enc_recon_weight, flow_recon_weight
    c.adaptive_loss_gamma, c.adaptive_loss_c = adaptive_loss_gamma, adaptive_loss_c
    c.use_imf, c.norm_p, c.norm_eps, c.aux_v_loss_weight = int(use_imf), norm_p, norm_eps, aux_v_loss_weight
    c.dispersive_loss_weight, c.dispersive_loss_tau = dispersive_loss_weight, dispersive_loss_tau
    c.freeze_encoder, c.freeze_decoder = int(freeze_encoder), int(freeze_decoder)
    c.enc_contrastive_weight, c.flow_contrastive_weight, c.temperature = enc_contrastive_weight, flow_contrastive_weight, temperature
    c.kl_weight = kl_weight
    return c


def compute_loss_meanflow(pw: MeanFlowPolicyHandle, cfg, obs_features: Tensor, actions: Tensor, t: Tensor, r: Tensor, *,
                          grads: Optional[MeanFlowPolicyHandle] = None, losses: Optional[Tensor] = None,
                          enc_noise: Optional[Tensor] = None, need_dobs: bool = False, grad_ready_events=None):
    """VitaPolicy.compute_loss with the MeanFlow matcher (+ backward when `grads` is given): vita_compute_loss_meanflow_f32.
    Returns losses (16,) indexed as _lib.MF_LOSS_NAMES (index 1 = meanflow_loss, or imf_loss + w * aux_v_loss); with need_dobs
    (and grads) the pair (losses, d loss / d obs_features).  No host sync."""
    lib = L.load()
    obs_features = _rows(obs_features, "compute_loss_meanflow: obs_features")
    actions = _f32(actions, "compute_loss_meanflow: actions").contiguous()
    t = _f32(t, "compute_loss_meanflow: t").contiguous()
    r = _f32(r, "compute_loss_meanflow: r").contiguous()
    B = obs_features.shape[0]
    if obs_features.shape[1] != pw.O or tuple(actions.shape) != (B, pw.T, pw.A) or tuple(t.shape) != (B,) or tuple(r.shape) != (B,):
        raise L.VitaHipError(f"compute_loss_meanflow: obs {tuple(obs_features.shape)} / actions {tuple(actions.shape)} / t {tuple(t.shape)} / "
                             f"r {tuple(r.shape)} do not match O={pw.O}, T={pw.T}, A={pw.A}")
    dev = obs_features.device
    if enc_noise is not None:
        if not pw.w.enc.variational:
            raise L.VitaHipError("compute_loss_meanflow: enc_noise given but the action encoder is not variational")
        enc_noise = _f32(enc_noise, "compute_loss_meanflow: enc_noise").contiguous()
        if tuple(enc_noise.shape) != (B, pw.D):
            raise L.VitaHipError(f"compute_loss_meanflow: enc_noise {tuple(enc_noise.shape)} is not (B, D) = ({B}, {pw.D})")
    if losses is None:
        losses = torch.empty(L.MF_LOSS_COUNT, dtype=torch.float32, device=dev)
    elif losses.numel() < L.MF_LOSS_COUNT:
        raise L.VitaHipError(f"compute_loss_meanflow: losses needs {L.MF_LOSS_COUNT} floats")
    nbytes = lib.vita_compute_loss_meanflow_workspace_bytes(C.byref(pw.w), C.byref(cfg), B)
    ws = workspace(dev, nbytes)
    if need_dobs and grads is None:
        raise L.VitaHipError("compute_loss_meanflow: need_dobs needs grads")
    ev = _event_handles(grad_ready_events, "compute_loss_meanflow")
    if ev is not None and len(ev) > 3:      # 4 + L events: per-section readiness (vita_meanflow_loss_config.grad_ready_layers)
        if len(ev) != 4 + pw.w.flow.L:
            raise L.VitaHipError(f"compute_loss_meanflow: {len(ev)} grad_ready events; pass 3, or 4 + L = {4 + pw.w.flow.L}")
        cfg = L.MeanFlowLossConfig.from_buffer_copy(cfg)
        cfg.grad_ready_layers = 1
    dobs = torch.empty_like(obs_features) if need_dobs else None
    L.check(lib.vita_compute_loss_meanflow_f32(C.byref(pw.w), C.byref(grads.w) if grads is not None else None, C.byref(cfg),
                                               obs_features.data_ptr(), _ld(obs_features), actions.data_ptr(), t.data_ptr(), r.data_ptr(),
                                               enc_noise.data_ptr() if enc_noise is not None else None, B, losses.data_ptr(),
                                               dobs.data_ptr() if dobs is not None else None, _ld(dobs) if dobs is not None else 0,
                                               ev,
                                               ws.data_ptr(), ws.numel(), _stream(obs_features)),
            "vita_compute_loss_meanflow_f32")
    return (losses, dobs) if need_dobs else losses


def action_encoder_forward(pw: PolicyHandle, actions: Tensor) -> Tensor:
    """CNNActionEncoder.forward / SimpleActionEncoder.forward (action_ae.py:318-327, 161-170).  For the variational
    encoder the result is the posterior's parameter rows (B, 2D) = [mean | logvar] (action_vae.py:224)."""
    lib = L.load()
    actions = _f32(actions, "action_encoder_forward").contiguous()
    B = actions.shape[0]
    width = pw.D * (2 if pw.w.enc.variational else 1)
    z = torch.empty((B, width), dtype=torch.float32, device=actions.device)
    nbytes = lib.vita_action_encoder_workspace_bytes(C.byref(pw.w.enc), B)
    ws = workspace(actions.device, nbytes)
    L.check(lib.vita_action_encoder_forward_f32(C.byref(pw.w.enc), actions.data_ptr(), z.data_ptr(), width, B, ws.data_ptr(),
                                                ws.numel(), _stream(actions)), "vita_action_encoder_forward_f32")
    return z


# --------------------------------------------------------------------------------------
# bf16-operand inference (fp32 accumulate / epilogue / residual stream)
# --------------------------------------------------------------------------------------
def pack_bf16(x: Tensor) -> Tensor:
    """fp32 (rows, cols) -> bf16 bits (rows, round_up(cols, 8)) as an int16 tensor, zero-padded (vita_pack_bf16_f32)."""
    lib = L.load()
    x = _rows(x, "pack_bf16: x")
    rows, cols = x.shape
    ld = (cols + 7) & ~7
    out = torch.empty((rows, ld), dtype=torch.int16, device=x.device)
    L.check(lib.vita_pack_bf16_f32(x.data_ptr(), _ld(x), out.data_ptr(), ld, rows, cols, _stream(x)), "vita_pack_bf16_f32")
    return out


def linear_bf16(xb: Tensor, wb: Tensor, b: Optional[Tensor] = None, act: int = L.ACT_NONE, *, n_out: Optional[int] = None,
                gate: Optional[Tensor] = None, gate_scalar: Optional[float] = None, resid: Optional[Tensor] = None,
                out_f32: bool = True, out_bf16: bool = False):
    """vita_linear_bf16 on packed operands (pack_bf16).  Returns (Y fp32 or None, Y bf16 bits or None)."""
    lib = L.load()
    M, Kp = xb.shape
    N = n_out if n_out is not None else wb.shape[0]
    if wb.shape[1] != Kp or xb.dtype != torch.int16 or wb.dtype != torch.int16:
        raise L.VitaHipError("linear_bf16: operands must be pack_bf16 outputs with the same padded K")
    y = torch.empty((M, N), dtype=torch.float32, device=xb.device) if out_f32 else None
    ldb = (N + 7) & ~7
    yb = torch.zeros((M, ldb), dtype=torch.int16, device=xb.device) if out_bf16 else None
    a = L.LinearArgs()
    a.X, a.ldx, a.W, a.ldw = xb.data_ptr(), Kp, wb.data_ptr(), Kp
    a.bias = b.contiguous().data_ptr() if b is not None else None
    a.Y, a.ldy = (y.data_ptr() if y is not None else None), N
    a.M, a.N, a.K, a.act = M, N, Kp, act
    keep = []
    if gate is not None:
        gate = gate.contiguous()
        keep.append(gate)
        a.gate, a.ld_gate = gate.data_ptr(), (0 if gate.dim() == 1 else N)
    if gate_scalar is not None:
        a.use_gate_scalar, a.gate_scalar = 1, float(gate_scalar)
    if resid is not None:
        resid = _rows(resid, "linear_bf16: resid")
        a.resid, a.ld_resid = resid.data_ptr(), _ld(resid)
    L.check(lib.vita_linear_bf16(C.byref(a), yb.data_ptr() if yb is not None else None, ldb, _stream(xb)), "vita_linear_bf16")
    return y, yb


def linear_bf16_tn(ab: Tensor, bb: Tensor, m: Optional[int] = None, n: Optional[int] = None, *, split: bool = True) -> Tensor:
    """C (m, n) fp32 = ab^T . bb for row-major bf16 bit tensors ab (R, >= m), bb (R, >= n) (pack_bf16 outputs): the weight
    gradient dW = dY^T X read from the operands as they lie in memory (vita_linear_bf16_tn).  split: let the reduction over R
    spread over workgroups (deterministic second pass)."""
    lib = L.load()
    if ab.dtype != torch.int16 or bb.dtype != torch.int16 or ab.dim() != 2 or bb.dim() != 2 or ab.shape[0] != bb.shape[0]:
        raise L.VitaHipError("linear_bf16_tn: operands must be pack_bf16 outputs with the same number of rows")
    if ab.stride(1) != 1 or bb.stride(1) != 1:
        raise L.VitaHipError("linear_bf16_tn: operands must have unit column stride")
    R = ab.shape[0]
    m = ab.shape[1] if m is None else m
    n = bb.shape[1] if n is None else n
    c = torch.empty((m, n), dtype=torch.float32, device=ab.device)
    ws = workspace(ab.device, lib.vita_splitk_workspace_bytes()) if split else None
    L.check(lib.vita_linear_bf16_tn(ab.data_ptr(), ab.stride(0), bb.data_ptr(), bb.stride(0), c.data_ptr(), n, m, n, R,
                                    ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream(ab)),
            "vita_linear_bf16_tn")
    return c


class Bf16Weights:
    """bf16 copies of the GEMM weights of a policy state dict (re-pack after every optimizer step that changes them)."""

    def __init__(self, sd: Mapping[str, Tensor], device):
        device = torch.device(device)
        self.t = {}
        g = lambda k: self.t.setdefault(k, pack_bf16(_prep(sd, k, device)))  # noqa: E731
        w = L.Bf16Weights()
        w.obs_w = g("obs_encoder.weight").data_ptr()
        w.in_w, w.out_w = g("flow_net.input_proj.weight").data_ptr(), g("flow_net.out_proj.weight").data_ptr()
        n = 0
        while f"flow_net.layers.{n}.mlp.fc1.weight" in sd:
            w.fc1_w[n] = g(f"flow_net.layers.{n}.mlp.fc1.weight").data_ptr()
            w.fc2_w[n] = g(f"flow_net.layers.{n}.mlp.fc2.weight").data_ptr()
            n += 1
        w.dec_in_w, w.dec_out_w = g("action_decoder.input_proj.weight").data_ptr(), g("action_decoder.output_proj.weight").data_ptr()
        n = 0
        while f"action_decoder.layers.{n}.fc1.weight" in sd:
            w.dec_fc1_w[n] = g(f"action_decoder.layers.{n}.fc1.weight").data_ptr()
            w.dec_fc2_w[n] = g(f"action_decoder.layers.{n}.fc2.weight").data_ptr()
            n += 1
        self.w = w


def generate_actions_bf16(obs_b: Tensor, fw: FlowNetHandle, dw: DecoderHandle, bw: Bf16Weights, obs_features: Tensor, num_steps: int,
                          return_latents: bool = False, out: Optional[Tensor] = None):
    """VitaPolicy.generate_actions with bf16 GEMM operands (vita_generate_actions_bf16)."""
    lib = L.load()
    obs_features = _rows(obs_features, "generate_actions_bf16: obs_features")
    obs_b = _f32(obs_b, "generate_actions_bf16: obs_b").contiguous()
    B, O = obs_features.shape
    if out is None:
        out = torch.empty((B, dw.OUT), dtype=torch.float32, device=obs_features.device)
    lat = torch.empty((B, fw.D), dtype=torch.float32, device=obs_features.device) if return_latents else None
    nbytes = lib.vita_generate_actions_bf16_workspace_bytes(C.byref(fw.w), C.byref(dw.w), O, B, num_steps)
    ws = workspace(obs_features.device, nbytes)
    L.check(lib.vita_generate_actions_bf16(obs_b.data_ptr(), O, C.byref(fw.w), C.byref(dw.w), C.byref(bw.w), obs_features.data_ptr(),
                                           _ld(obs_features), out.data_ptr(), dw.OUT, lat.data_ptr() if lat is not None else None,
                                           B, num_steps, ws.data_ptr(), ws.numel(), _stream(obs_features)),
            "vita_generate_actions_bf16")
    return (out, lat) if return_latents else out
