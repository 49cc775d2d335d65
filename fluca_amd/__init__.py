"""fluca_amd -- MI355X-native pressure-Poisson / IBM path of Fluca behind a C-ABI (include/fluca_hip.h).

This package is only the host-side plumbing (ctypes + torch device memory) around libflucahip.so.  There is no CPU
fallback: if the HIP library is missing or fails to load, importing `fluca_amd.capi` raises.
"""
__version__ = "0.1"
