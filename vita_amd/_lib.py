"""ctypes binding of libvita_hip.so (the C ABI declared in include/vita_hip.h).

There is no fallback: if the shared library is missing or fails to load, importing the
compute entry points raises — the product path never silently runs anything else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvita_hip.so")
MAX_LAYERS = 16

ACT_NONE, ACT_RELU, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SILU, ACT_MISH = range(6)
PRO_NONE, PRO_LN_MOD, PRO_LN_AFFINE = range(3)

c_fp = C.c_void_p  # device pointers travel as integers


class LinearArgs(C.Structure):
    _fields_ = [
        ("X", c_fp), ("ldx", C.c_int64),
        ("W", c_fp), ("ldw", C.c_int64),
        ("bias", c_fp),
        ("Y", c_fp), ("ldy", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("act", C.c_int32), ("pro", C.c_int32), ("ln_eps", C.c_float),
        ("pro_mul", c_fp), ("ld_pro_mul", C.c_int64),
        ("pro_add", c_fp), ("ld_pro_add", C.c_int64),
        ("gate", c_fp), ("ld_gate", C.c_int64),
        ("use_gate_scalar", C.c_int32), ("gate_scalar", C.c_float),
        ("resid", c_fp), ("ld_resid", C.c_int64),
        ("preact", c_fp), ("ld_preact", C.c_int64),
        ("a_col", C.c_int32), ("b_col", C.c_int32),
        ("act_grad_z", c_fp), ("ld_act_grad_z", C.c_int64),
        ("splitk_ws", c_fp), ("splitk_ws_bytes", C.c_size_t),
    ]


_PA = c_fp * MAX_LAYERS


class FlowNetWeights(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("H", C.c_int32), ("FF", C.c_int32), ("L", C.c_int32), ("TE", C.c_int32),
        ("in_w", c_fp), ("in_b", c_fp),
        ("te1_w", c_fp), ("te1_b", c_fp),
        ("te3_w", c_fp), ("te3_b", c_fp),
        ("fc1_w", _PA), ("fc1_b", _PA),
        ("fc2_w", _PA), ("fc2_b", _PA),
        ("mod_w", _PA), ("mod_b", _PA),
        ("norm_w", c_fp), ("norm_b", c_fp),
        ("out_w", c_fp), ("out_b", c_fp),
    ]


class MeanFlowWeights(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("H", C.c_int32), ("FF", C.c_int32), ("L", C.c_int32),
        ("t1_w", c_fp), ("t1_b", c_fp), ("t2_w", c_fp), ("t2_b", c_fp),
        ("h1_w", c_fp), ("h1_b", c_fp), ("h2_w", c_fp), ("h2_b", c_fp),
        ("in_w", c_fp), ("in_b", c_fp),
        ("fc1_w", _PA), ("fc1_b", _PA),
        ("fc2_w", _PA), ("fc2_b", _PA),
        ("u_w", c_fp), ("u_b", c_fp), ("v_w", c_fp), ("v_b", c_fp),
    ]


class DecoderWeights(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("HD", C.c_int32), ("NL", C.c_int32), ("OUT", C.c_int32),
        ("in_w", c_fp), ("in_b", c_fp),
        ("fc1_w", _PA), ("fc1_b", _PA),
        ("fc2_w", _PA), ("fc2_b", _PA),
        ("out_w", c_fp), ("out_b", c_fp),
    ]


class EncoderWeights(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("A", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("NL", C.c_int32), ("D", C.c_int32),
        ("w1", _PA), ("b1", _PA), ("w2", _PA), ("b2", _PA),
        ("proj_w", c_fp), ("proj_b", c_fp),
        ("variational", C.c_int32),
    ]


class PolicyWeights(C.Structure):
    _fields_ = [
        ("O", C.c_int32),
        ("obs_w", c_fp), ("obs_b", c_fp),
        ("flow", FlowNetWeights),
        ("enc", EncoderWeights),
        ("dec", DecoderWeights),
    ]


class LossConfig(C.Structure):
    _fields_ = [
        ("num_steps", C.c_int32), ("recon_l1", C.c_int32), ("decode_flow_latents", C.c_int32),
        ("consistency_weight", C.c_float), ("enc_recon_weight", C.c_float), ("flow_recon_weight", C.c_float),
        ("enc_contrastive_weight", C.c_float), ("flow_contrastive_weight", C.c_float), ("temperature", C.c_float),
        ("freeze_encoder", C.c_int32), ("freeze_decoder", C.c_int32),
        ("kl_weight", C.c_float), ("gemm_bf16", C.c_int32),
        ("fm_kind", C.c_int32), ("fm_sigma", C.c_float), ("side_streams", C.c_int32), ("grad_ready_layers", C.c_int32),
        ("external_decoder", C.c_int32), ("resume_backward", C.c_int32),
    ]


class MeanFlowPolicyWeights(C.Structure):
    _fields_ = [
        ("O", C.c_int32),
        ("obs_w", c_fp), ("obs_b", c_fp),
        ("flow", MeanFlowWeights),
        ("enc", EncoderWeights),
        ("dec", DecoderWeights),
    ]


class MeanFlowLossConfig(C.Structure):
    _fields_ = [
        ("num_steps", C.c_int32), ("recon_l1", C.c_int32), ("decode_flow_latents", C.c_int32),
        ("consistency_weight", C.c_float), ("enc_recon_weight", C.c_float), ("flow_recon_weight", C.c_float),
        ("adaptive_loss_gamma", C.c_float), ("adaptive_loss_c", C.c_float),
        ("use_imf", C.c_int32), ("norm_p", C.c_float), ("norm_eps", C.c_float), ("aux_v_loss_weight", C.c_float),
        ("dispersive_loss_weight", C.c_float), ("dispersive_loss_tau", C.c_float),
        ("freeze_encoder", C.c_int32), ("freeze_decoder", C.c_int32),
        ("enc_contrastive_weight", C.c_float), ("flow_contrastive_weight", C.c_float), ("temperature", C.c_float),
        ("kl_weight", C.c_float), ("grad_ready_layers", C.c_int32),
    ]


class Bf16Weights(C.Structure):
    _fields_ = [
        ("obs_w", c_fp), ("in_w", c_fp), ("out_w", c_fp),
        ("fc1_w", _PA), ("fc2_w", _PA),
        ("dec_in_w", c_fp), ("dec_out_w", c_fp),
        ("dec_fc1_w", _PA), ("dec_fc2_w", _PA),
    ]


ENC_CNN, ENC_SIMPLE, ENC_EXTERNAL = 0, 1, 2
FM_CONDITIONAL, FM_TARGET, FM_SCHRODINGER = 0, 1, 2
LOSS_NAMES = ("loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss", "enc_contrastive_loss",
              "flow_contrastive_loss", "action_kl_loss")
LOSS_COUNT = 8
MF_LOSS_NAMES = ("loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss", "imf_loss", "aux_v_loss",
                 "imf_mse", "aux_v_mse", "dispersive_loss", "flow_loss_base", "enc_contrastive_loss", "flow_contrastive_loss", "action_kl_loss")
MF_LOSS_COUNT = 16

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks the header against
SIGNATURES = {
    "vita_abi_version": (C.c_int, []),
    "vita_last_error": (C.c_char_p, []),
    "vita_linear_f32": (C.c_int, [C.POINTER(LinearArgs), c_fp]),
    "vita_splitk_workspace_bytes": (C.c_size_t, []),
    "vita_debug_last_gemm_config": (C.c_char_p, []),
    "vita_flownet_forward_workspace_bytes": (C.c_size_t, [C.POINTER(FlowNetWeights), C.c_int32]),
    "vita_flownet_forward_f32": (C.c_int, [C.POINTER(FlowNetWeights), c_fp, C.c_int64, c_fp, c_fp, C.c_int64,
                                           C.c_int32, c_fp, C.c_size_t, c_fp]),
    "vita_ode_sample_workspace_bytes": (C.c_size_t, [C.POINTER(FlowNetWeights), C.c_int32, C.c_int32]),
    "vita_ode_sample_f32": (C.c_int, [C.POINTER(FlowNetWeights), c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32,
                                      C.c_int32, c_fp, c_fp, C.c_size_t, c_fp]),
    "vita_ode_sample_set_persistent": (C.c_int, [C.c_int32]),
    "vita_ode_sample_takes_persistent": (C.c_int, [C.POINTER(FlowNetWeights), C.c_int32]),
    "vita_debug_ode_stamps": (C.c_int, [c_fp]),
    "vita_ode_sample_status": (C.c_int, [C.POINTER(FlowNetWeights), C.c_int32, C.c_int32, c_fp, C.POINTER(C.c_int32)]),
    "vita_generate_actions_status": (C.c_int, [C.POINTER(FlowNetWeights), C.c_int32, C.c_int32, c_fp, C.POINTER(C.c_int32)]),
    "vita_generate_actions_noalloc_status": (C.c_int, [C.POINTER(FlowNetWeights), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "vita_generate_actions_bf16_status": (C.c_int, [C.POINTER(FlowNetWeights), C.c_int32, C.c_int32, C.c_int32, c_fp, C.POINTER(C.c_int32)]),
    "vita_debug_set_spin_limit": (C.c_int, [C.c_uint32]),
    "vita_decoder_forward_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderWeights), C.c_int32]),
    "vita_decoder_forward_f32": (C.c_int, [C.POINTER(DecoderWeights), c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32,
                                           c_fp, C.c_size_t, c_fp]),
    "vita_generate_actions_workspace_bytes": (C.c_size_t, [C.POINTER(FlowNetWeights), C.POINTER(DecoderWeights),
                                                           C.c_int32, C.c_int32]),
    "vita_generate_actions_f32": (C.c_int, [c_fp, c_fp, C.c_int32, C.POINTER(FlowNetWeights), C.POINTER(DecoderWeights),
                                            c_fp, C.c_int64, c_fp, C.c_int64, c_fp, C.c_int32, C.c_int32,
                                            c_fp, C.c_size_t, c_fp]),
    "vita_noalloc_scratch_bytes": (C.c_size_t, []),
    "vita_noalloc_init": (C.c_int, []),
    "vita_generate_actions_noalloc_need_bytes": (C.c_size_t, [C.POINTER(FlowNetWeights), C.POINTER(DecoderWeights),
                                                              C.c_int32, C.c_int32]),
    "vita_generate_actions_noalloc_f32": (C.c_int, [c_fp, c_fp, C.c_int32, C.POINTER(FlowNetWeights), C.POINTER(DecoderWeights),
                                                    c_fp, C.c_int64, c_fp, C.c_int64, c_fp, C.c_int32, C.c_int32, c_fp]),
    "vita_loss_partials_count": (C.c_int32, []),
    "vita_mse_loss_fwd_f32": (C.c_int, [c_fp, c_fp, C.c_int64, c_fp, c_fp, c_fp]),
    "vita_mse_loss_bwd_f32": (C.c_int, [c_fp, c_fp, C.c_int64, c_fp, c_fp, c_fp, C.c_int32, c_fp]),
    "vita_l1_loss_fwd_f32": (C.c_int, [c_fp, c_fp, C.c_int64, c_fp, c_fp, c_fp]),
    "vita_l1_loss_bwd_f32": (C.c_int, [c_fp, c_fp, C.c_int64, c_fp, c_fp, c_fp, C.c_int32, c_fp]),
    "vita_cfm_interpolate_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int32, C.c_int32, c_fp]),
    "vita_cfm_interpolate_bwd_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_act_fwd_f32": (C.c_int, [c_fp, c_fp, C.c_int64, C.c_int32, c_fp]),
    "vita_act_bwd_f32": (C.c_int, [c_fp, c_fp, c_fp, C.c_int64, C.c_int32, c_fp]),
    "vita_gate_bwd_f32": (C.c_int, [c_fp, c_fp, c_fp, C.c_int64, C.c_float, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int32, c_fp]),
    "vita_layernorm_modulate_bwd_f32": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32,
                                                  C.c_float, C.c_int32, c_fp, C.c_int64, C.c_int32, c_fp, c_fp, C.c_int64, c_fp]),
    "vita_compute_loss_workspace_bytes": (C.c_size_t, [C.POINTER(PolicyWeights), C.POINTER(LossConfig), C.c_int32]),
    "vita_compute_loss_f32": (C.c_int, [C.POINTER(PolicyWeights), C.POINTER(PolicyWeights), C.POINTER(LossConfig), c_fp, C.c_int64,
                                        c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int32, c_fp, c_fp, C.c_int64, C.POINTER(c_fp), c_fp,
                                        C.c_size_t, c_fp]),
    "vita_fm_interpolate_f32": (C.c_int, [C.c_int32, C.c_double, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int32, C.c_int32, c_fp]),
    "vita_fm_interpolate_bwd_f32": (C.c_int, [C.c_int32, C.c_double, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int32, C.c_int32, C.c_int32,
                                              c_fp]),
    "vita_action_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderWeights), C.c_int32]),
    "vita_action_encoder_forward_f32": (C.c_int, [C.POINTER(EncoderWeights), c_fp, c_fp, C.c_int64, C.c_int32, c_fp, C.c_size_t,
                                                  c_fp]),
    "vita_gather_window_f32": (C.c_int, [c_fp, C.c_int64, C.c_int32, c_fp, C.c_int32, c_fp, C.c_int32, c_fp, c_fp, c_fp, C.c_int64, C.c_int64,
                                         c_fp, c_fp, c_fp, c_fp]),
    "vita_gather_window_u8_nchw_f32": (C.c_int, [c_fp, C.c_int32, C.c_int32, C.c_int32, c_fp, C.c_int32, c_fp, C.c_int32, c_fp, c_fp, c_fp,
                                                 C.c_int64, C.c_int64, c_fp, c_fp, c_fp, c_fp]),
    "vita_channel_affine_relu_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_channel_affine_relu_bwd_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_affine_relu_maxpool3s2_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_affine_relu_maxpool3s2_bwd_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_colsum_partials_count": (C.c_int64, [C.c_int32]),
    "vita_colsum_f32": (C.c_int, [c_fp, C.c_int64, C.c_int32, C.c_int32, c_fp, c_fp, C.c_int32, c_fp]),
    "vita_im2col_k5s2_f32": (C.c_int, [c_fp, c_fp, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_col2im_k5s2_f32": (C.c_int, [c_fp, c_fp, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_nlc_ncl_f32": (C.c_int, [c_fp, c_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_fp]),
    "vita_sumsq_f32": (C.c_int, [c_fp, C.c_int64, c_fp, c_fp, c_fp]),
    "vita_adamw_ema_step_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int64, c_fp, C.c_float, C.c_float, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_double, c_fp]),
    "vita_pack_bf16_f32": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, C.c_int64, C.c_int32, c_fp]),
    "vita_linear_bf16": (C.c_int, [C.POINTER(LinearArgs), c_fp, C.c_int64, c_fp]),
    "vita_linear_bf16_tn": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_fp, C.c_size_t, c_fp]),
    "vita_generate_actions_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(FlowNetWeights), C.POINTER(DecoderWeights), C.c_int32,
                                                                C.c_int32, C.c_int32]),
    "vita_generate_actions_bf16": (C.c_int, [c_fp, C.c_int32, C.POINTER(FlowNetWeights), C.POINTER(DecoderWeights),
                                             C.POINTER(Bf16Weights), c_fp, C.c_int64, c_fp, C.c_int64, c_fp, C.c_int32, C.c_int32,
                                             c_fp, C.c_size_t, c_fp]),
    "vita_contrastive_loss_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "vita_contrastive_loss_f32": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_float, c_fp, c_fp, c_fp, C.c_int64,
                                            c_fp, C.c_int64, c_fp, C.c_size_t, c_fp]),
    "vita_gaussian_kl_fwd_f32": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, c_fp, c_fp, c_fp]),
    "vita_gaussian_kl_bwd_f32": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, c_fp, c_fp, C.c_int64, c_fp, C.c_int64,
                                           c_fp]),
    "vita_consistency_sample_workspace_bytes": (C.c_size_t, [C.POINTER(FlowNetWeights), C.c_int32, C.c_int32]),
    "vita_consistency_sample_f32": (C.c_int, [C.POINTER(FlowNetWeights), c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                              C.c_float, C.c_float, c_fp, c_fp, C.c_size_t, c_fp]),
    "vita_meanflow_forward_workspace_bytes": (C.c_size_t, [C.POINTER(MeanFlowWeights), C.c_int32]),
    "vita_meanflow_forward_f32": (C.c_int, [C.POINTER(MeanFlowWeights), c_fp, C.c_int64, c_fp, c_fp, c_fp, C.c_int64, c_fp, C.c_int64,
                                            C.c_int32, c_fp, C.c_size_t, c_fp]),
    "vita_meanflow_sample_workspace_bytes": (C.c_size_t, [C.POINTER(MeanFlowWeights), C.c_int32, C.c_int32]),
    "vita_meanflow_sample_f32": (C.c_int, [C.POINTER(MeanFlowWeights), c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, c_fp,
                                           C.c_size_t, c_fp]),
    "vita_compute_loss_meanflow_workspace_bytes": (C.c_size_t, [C.POINTER(MeanFlowPolicyWeights), C.POINTER(MeanFlowLossConfig), C.c_int32]),
    "vita_compute_loss_meanflow_f32": (C.c_int, [C.POINTER(MeanFlowPolicyWeights), C.POINTER(MeanFlowPolicyWeights),
                                                 C.POINTER(MeanFlowLossConfig), c_fp, C.c_int64, c_fp, c_fp, c_fp, c_fp, C.c_int32, c_fp,
                                                 c_fp, C.c_int64, C.POINTER(c_fp), c_fp, C.c_size_t, c_fp]),
    "vita_normalize_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int64, c_fp]),
    "vita_unnormalize_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int32, C.c_int64, c_fp]),
    "vita_layernorm_modulate_f32": (C.c_int, [c_fp, C.c_int64, c_fp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                              c_fp, C.c_int64, c_fp, C.c_int64, c_fp]),
    "vita_sinusoidal_embed_f32": (C.c_int, [c_fp, c_fp, C.c_int64, C.c_int32, C.c_int32, c_fp]),
}

_lib = None
_bad_fork = False


class VitaHipError(RuntimeError):
    pass


_gpu_live_before_fork = False


def _before_fork():
    """In the parent, just before fork(): does this process hold a HIP context that the library may have used?  (Asked here
    because torch.cuda.is_initialized() answers False inside a forked child.)"""
    global _gpu_live_before_fork
    try:
        import sys
        torch = sys.modules.get("torch")
        _gpu_live_before_fork = bool(_lib is not None and torch is not None and torch.cuda.is_initialized())
    except Exception:
        _gpu_live_before_fork = False


def _after_fork_in_child():
    """HIP, like CUDA, cannot be used in a fork()ed child of a process that had initialised it (the runtime's threads and
    its queues belong to the parent).  torch raises for its own calls ("Cannot re-initialize CUDA in forked subprocess");
    calls that go straight to HIP through this ctypes binding bypass that check and HUNG intermittently on the GPU box
    (round 1, tests/test_gpu_00_dp.py).  Mark the child so that load() fails loudly instead."""
    global _bad_fork
    if _gpu_live_before_fork:
        _bad_fork = True


os.register_at_fork(before=_before_fork, after_in_child=_after_fork_in_child)


def load() -> C.CDLL:
    """Load libvita_hip.so and bind every symbol the header declares.  Raises if the
    library has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _bad_fork:
        raise VitaHipError("this process was fork()ed from one that had initialised the GPU: HIP cannot be used here. Start worker "
                           "processes with the 'spawn' method (torch.multiprocessing.get_context('spawn')), or fork before any GPU call")
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitaHipError(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). "
            "vita_amd has no CPU or PyTorch fallback on purpose.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch; let it surface
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vita_last_error()
        raise VitaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
