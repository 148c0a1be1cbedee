#!/bin/bash
# Debug-library variant built from the sources of a git revision (default HEAD), for interleaved A/B runs against the working tree's debug build:
#   bash tools/build_debug_head.sh [rev = HEAD] [tag = 1]   ->  tools/_build/libpwr_hip_dbg_DPWR_VARIANT_OLD_<tag>.so
#   python tools/ab_step.py - PWR_DBGLIB=libpwr_hip_dbg_DPWR_VARIANT_OLD_<tag>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); REV=${1:-HEAD}; TAG=${2:-1}
T=$(mktemp -d); cd $R
cp -r pixelwiseregression_amd/csrc/*.hip pixelwiseregression_amd/csrc/*.h pixelwiseregression_amd/csrc/*.inc include/pwr.h include/pwr_debug.h $T/
restore() { cp $T/*.hip $T/*.inc pixelwiseregression_amd/csrc/ 2>/dev/null; cp $T/conv_common.h $T/pwr_common.h pixelwiseregression_amd/csrc/; cp $T/pwr.h $T/pwr_debug.h include/; rm -rf $T; }
trap restore EXIT
git checkout -q $REV -- pixelwiseregression_amd/csrc include/pwr.h include/pwr_debug.h
rm -f tools/_build/libpwr_hip_dbg_DPWR_VARIANT_OLD_$TAG.so*
python tools/build_debug.py -DPWR_VARIANT_OLD=$TAG | tail -1
git checkout -q HEAD -- pixelwiseregression_amd/csrc include/pwr.h include/pwr_debug.h 2>/dev/null || true
