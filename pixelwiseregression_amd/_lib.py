"""ctypes binding of libpwr_hip.so (C ABI: include/pwr.h).

There is NO fallback: if the library is missing or a call fails this raises.  A CPU/ATen fallback
would silently void the parity claims, so the product path refuses to run without the HIP kernels.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpwr_hip.so")
ABI_VERSION = 1

_lib = None

c_f32p = ctypes.c_void_p
c_int = ctypes.c_int
c_vp = ctypes.c_void_p

# name -> argtypes (restype is always int); mirrors include/pwr.h one to one
SIGNATURES = {
    "pwr_abi_version": [],
    "pwr_decode_fwd": [c_vp] * 7 + [c_int] * 4 + [c_vp],
    "pwr_decode_bwd": [c_vp] * 13 + [c_int] * 4 + [c_vp],
    "pwr_decode_gw_reduce": [c_vp, c_vp, c_int, c_int, c_int, c_vp],
}


class PwrError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PwrError("libpwr_hip.so not built (%s). Run `python -m pixelwiseregression_amd.build` "
                           "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)       # AttributeError if the symbol is missing -> loud
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        if l.pwr_abi_version() != ABI_VERSION:
            raise PwrError("libpwr_hip.so ABI %d != binding %d: rebuild" % (l.pwr_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        raise PwrError("%s failed with code %d" % (what, code))


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "pwr kernels need contiguous tensors"
    return t.data_ptr()


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
