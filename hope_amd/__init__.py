"""hope_amd — MI355X-native HoPE horizontal-plane extraction (host-side binding).

The compute lives in hope_amd/csrc (hand-written HIP for gfx950 behind the C ABI of
include/hope_gpu.h).  This package is a thin ctypes mirror; it has no CPU fallback and
raises if the HIP library has not been built.
"""
from .capi import (HOPE_EMPTY, HOPE_MODE_OFFLINE, HOPE_MODE_RT, HOPE_NO_NORMAL_FIT, HOPE_OK,  # noqa: F401
                   HIP_LIB_PATH, HopeContext, HopeError, HopeLib)
