"""Build libpwr_hip_rbplain.so: the library with the fused ResBlock kernels reading threadIdx.x directly (PWR_RB_OPAQUE_TID=0), for the
same-box A/B of csrc/resblock_small.hip's opaque work-item id.  Load it with PWR_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd.build import build, HERE
build(extra_flags=["-DPWR_RB_OPAQUE_TID=0"], lib=os.path.join(HERE, "libpwr_hip_rbplain.so"), obj=os.path.join(HERE, "csrc", "_obj_rbplain"))
