"""Build libpwr_hip_w3dbg.so: the library with the timing-by-elimination variants of conv_wgrad3_kernel (PWR_WGRAD3_DBG=1|2|3|4|16|19|48|51,
see csrc/conv_mfma.hip; their RESULTS ARE WRONG by construction).  Load it with PWR_LIB; tools/bench_kernels.py wgrad times it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd.build import build, HERE
build(extra_flags=["-DPWR_WGRAD3_DBG_BUILD"], lib=os.path.join(HERE, "libpwr_hip_w3dbg.so"), obj=os.path.join(HERE, "csrc", "_obj_w3dbg"))
