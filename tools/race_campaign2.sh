#!/bin/bash
# Like race_campaign.sh, variant under test: the library built with -fno-slp-vectorize (no compiler-packed f32 math), join-once.
out=${1:-gpurun_out/race2}; probe=${2:-15000}; var=${3:-40000}
mkdir -p $out
RACE_DUMP=$out/dump_base.pt PWR_JOIN_ONCE=1 python tools/race_hunt.py $probe > $out/hunt_base.txt 2>&1
head -3 $out/hunt_base.txt | tail -2
if grep -q "different gradient: 0 " $out/hunt_base.txt; then echo "box does not reproduce in $probe steps"; exit 0; fi
RACE_DUMP=$out/dump_noslp.pt PWR_JOIN_ONCE=1 PWR_LIB=$PWD/pixelwiseregression_amd/libpwr_hip_noslp.so python tools/race_hunt.py $var > $out/hunt_noslp.txt 2>&1; head -2 $out/hunt_noslp.txt | tail -1
RACE_DUMP=$out/dump_base2.pt PWR_JOIN_ONCE=1 python tools/race_hunt.py $probe > $out/hunt_base2.txt 2>&1; head -2 $out/hunt_base2.txt | tail -1
