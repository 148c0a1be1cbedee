"""Build the DEBUG variant of the library: tools/_build/libpwr_hip_dbg.so (-DPWR_DEBUG_BUILD: the experiment switches of the kernels
and the engine read their PWR_* environment variables, the debugging entry points of include/pwr_debug.h exist).  The shipped
pixelwiseregression_amd/libpwr_hip.so has one configuration and reads no experiment variable; measurement scripts load this build
through tools/dbglib.py.

    python tools/build_debug.py [-DNAME=VALUE ...] [--force]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd.build import build

if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith(("-D", "-f"))]
    print(build(force="--force" in sys.argv, debug=True, extra_flags=extra))
