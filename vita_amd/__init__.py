"""vita_amd — MI355X-native kernels behind VITA's vision-to-action flow-matching hot path.

`vita_amd.ops`    functional operators over torch HIP tensors (ctypes -> libvita_hip.so)
`vita_amd.policy` VitaPolicy-compatible host mirror (same names / state-dict keys as the reference)
"""
__all__ = ["ops", "_lib"]
