"""CPU-side ABI checks: the shared library loads without a GPU and exports exactly the entry
points include/vita_hip.h declares; the ctypes table in vita_amd/_lib.py covers all of them."""
import ctypes
import os
import re

from vita_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "vita_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vita_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_ctypes_binds():
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libvita_hip.so missing: run `make` / __graft_entry__.build()"
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.vita_abi_version() == 2
    assert lib.vita_loss_partials_count() == 1024


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof of the POD argument structs as a C compiler sees the header, against the
    ctypes mirrors (gcc is part of the image; the header is plain C)."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "vita_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vita_linear_args), '
        'sizeof(vita_flownet_weights), sizeof(vita_decoder_weights), offsetof(vita_linear_args, resid), '
        'offsetof(vita_linear_args, gate_scalar), offsetof(vita_flownet_weights, out_b), '
        'offsetof(vita_decoder_weights, out_w), sizeof(vita_encoder_weights), sizeof(vita_policy_weights), '
        'sizeof(vita_loss_config), offsetof(vita_policy_weights, dec), offsetof(vita_encoder_weights, proj_w), '
        'offsetof(vita_linear_args, act_grad_z), sizeof(vita_meanflow_weights), offsetof(vita_meanflow_weights, u_w), '
        'sizeof(vita_meanflow_loss_config), offsetof(vita_meanflow_loss_config, kl_weight), offsetof(vita_meanflow_loss_config, use_imf), '
        'sizeof(vita_meanflow_policy_weights), offsetof(vita_meanflow_policy_weights, dec), sizeof(vita_bf16_weights), '
        'offsetof(vita_bf16_weights, dec_fc2_w), offsetof(vita_loss_config, resume_backward)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.LinearArgs), ctypes.sizeof(_lib.FlowNetWeights), ctypes.sizeof(_lib.DecoderWeights),
            _lib.LinearArgs.resid.offset, _lib.LinearArgs.gate_scalar.offset, _lib.FlowNetWeights.out_b.offset,
            _lib.DecoderWeights.out_w.offset, ctypes.sizeof(_lib.EncoderWeights), ctypes.sizeof(_lib.PolicyWeights),
            ctypes.sizeof(_lib.LossConfig), _lib.PolicyWeights.dec.offset, _lib.EncoderWeights.proj_w.offset,
            _lib.LinearArgs.act_grad_z.offset, ctypes.sizeof(_lib.MeanFlowWeights), _lib.MeanFlowWeights.u_w.offset,
            ctypes.sizeof(_lib.MeanFlowLossConfig), _lib.MeanFlowLossConfig.kl_weight.offset, _lib.MeanFlowLossConfig.use_imf.offset,
            ctypes.sizeof(_lib.MeanFlowPolicyWeights), _lib.MeanFlowPolicyWeights.dec.offset, ctypes.sizeof(_lib.Bf16Weights),
            _lib.Bf16Weights.dec_fc2_w.offset, _lib.LossConfig.resume_backward.offset]
    assert got == want


def test_host_side_validation_needs_no_gpu():
    """Argument validation happens on the host before any launch, so bad calls fail cleanly
    even here; workspace size queries are pure host arithmetic."""
    lib = _lib.load()
    a = _lib.LinearArgs()
    assert lib.vita_linear_f32(ctypes.byref(a), None) == -1
    assert b"non-null" in lib.vita_last_error()
    w = _lib.FlowNetWeights()
    w.D, w.H, w.FF, w.L, w.TE = 512, 512, 2048, 4, 256
    n6 = lib.vita_ode_sample_workspace_bytes(ctypes.byref(w), 4096, 6)
    n100 = lib.vita_ode_sample_workspace_bytes(ctypes.byref(w), 4096, 100)
    assert n6 >= 4096 * (512 + 2048) * 4 and n100 > n6
    assert lib.vita_ode_sample_workspace_bytes(ctypes.byref(w), 0, 6) == 0
    assert lib.vita_ode_sample_f32(ctypes.byref(w), None, 512, None, 512, 4, 0, None, None, 0, None) == -1
