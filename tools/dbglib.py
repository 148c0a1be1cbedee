"""Make pixelwiseregression_amd load the DEBUG build of the library (tools/_build/libpwr_hip_dbg.so, tools/build_debug.py) and bind
the debugging entry points of include/pwr_debug.h as well.  Import this BEFORE anything calls _lib.lib():

    import dbglib            # (tools/ on sys.path)  -> the experiment switches (PWR_* environment variables) are live
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from pixelwiseregression_amd import _lib

DBG = os.path.join(ROOT, "tools", "_build", os.environ.get("PWR_DBGLIB", "libpwr_hip_dbg.so"))    # PWR_DBGLIB: a variant built with extra flags
if _lib._lib is not None:
    raise RuntimeError("dbglib must be imported before the product library is loaded")
if not os.path.exists(DBG):
    raise RuntimeError("debug library not built: python tools/build_debug.py")
_lib.LIB_PATH = DBG
_lib.SIGNATURES = _lib._parse_header(["pwr.h", "pwr_debug.h"])
