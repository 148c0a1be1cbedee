"""Build libpwr_hip_slp.so: the library WITH the SLP vectoriser (= the round-1 build flags), used only as the probe of
tools/race_campaign*.sh / soak_campaign.sh to classify a box (does it show the round-1 events?).  Load it with PWR_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd.build import build, HERE
build(extra_flags=["-fslp-vectorize"], lib=os.path.join(HERE, "libpwr_hip_slp.so"), obj=os.path.join(HERE, "csrc", "_obj_slp"))
