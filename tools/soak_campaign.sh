#!/bin/bash
# Soak of the DEFAULT configuration (product library, join per segment) on a box classified first: a short probe with the
# SLP-vectorised library in the join-once mode tells whether this box shows the round-1 events at all.
out=${1:-gpurun_out/soak}; probe=${2:-15000}; n=${3:-60000}
mkdir -p $out
PWR_JOIN_ONCE=1 PWR_LIB=$PWD/pixelwiseregression_amd/libpwr_hip_slp.so python tools/race_hunt.py $probe 2>&1 | grep "join_once" > $out/probe_slp.txt; cat $out/probe_slp.txt
python tools/determinism.py $n 2>&1 | grep -v amdgpu.ids > $out/soak_default.txt; head -3 $out/soak_default.txt
