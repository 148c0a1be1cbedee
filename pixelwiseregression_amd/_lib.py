"""ctypes binding of libpwr_hip.so (C ABI: include/pwr.h).

There is NO fallback: if the library is missing or a call fails this raises.  A CPU/ATen fallback
would silently void the parity claims, so the product path refuses to run without the HIP kernels.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpwr_hip.so")     # the one product library (tools/dbglib.py swaps in the debug build for measurements)
ABI_VERSION = 7
_HEADERS = ["pwr.h"]

_lib = None

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "long long": ctypes.c_longlong,
           "void": None, "void*": ctypes.c_void_p, "const char*": ctypes.c_char_p}


def _parse_header(names=None):
    """name -> (restype, [argtypes]) parsed from include/pwr.h, so the binding cannot drift from the header."""
    import re
    txt = ""
    for h in (names or _HEADERS):
        txt += re.sub(r"/\*.*?\*/", "", open(os.path.join(os.path.dirname(_HERE), "include", h)).read(), flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int|size_t|void\*|void|long long|const char\*)\s+(pwr_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    t = a.replace("const ", "").rsplit(" ", 1)[0].strip()
                    argtypes.append(_CTYPES[t])
        sigs[name] = (_CTYPES[ret], argtypes)
    return sigs


SIGNATURES = _parse_header()


class PwrError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PwrError("libpwr_hip.so not built (%s). Run `python -m pixelwiseregression_amd.build` "
                           "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        if l.pwr_abi_version() != ABI_VERSION:
            raise PwrError("%s has ABI %d, this binding needs %d: rebuild (python -m pixelwiseregression_amd.build)"
                           % (LIB_PATH, l.pwr_abi_version(), ABI_VERSION))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(l, name)       # AttributeError if the symbol is missing -> loud
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        msg = ""
        try:
            msg = (lib().pwr_last_error() or b"").decode()
        except Exception:
            pass
        raise PwrError("%s failed with code %d %s" % (what, code, msg))


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "pwr kernels need contiguous tensors"
    return t.data_ptr()


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
