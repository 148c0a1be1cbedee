#!/bin/bash
# One box of the pool: does the rare non-reproducible step show here?  If it does, capture it (raw decoder buffers) and run the
# discriminating variants on the SAME box; if not, leave quickly (the event is box dependent).
#   tools/race_campaign.sh <outdir> [probe_steps] [variant_steps]
out=${1:-gpurun_out/race}; probe=${2:-15000}; var=${3:-30000}
mkdir -p $out
RACE_DUMP=$out/dump_base.pt PWR_JOIN_ONCE=1 python tools/race_hunt.py $probe > $out/hunt_base.txt 2>&1
head -3 $out/hunt_base.txt | tail -2
if grep -q "different gradient: 0 " $out/hunt_base.txt; then echo "box does not reproduce in $probe steps"; exit 0; fi
RACE_DUMP=$out/dump_base2.pt PWR_JOIN_ONCE=1 python tools/race_hunt.py $var > $out/hunt_base2.txt 2>&1; head -2 $out/hunt_base2.txt | tail -1
RACE_DUMP=$out/dump_sc1.pt PWR_JOIN_ONCE=1 PWR_DEC_SCALAR_SC1=1 python tools/race_hunt.py $var > $out/hunt_sc1.txt 2>&1; head -2 $out/hunt_sc1.txt | tail -1
RACE_DUMP=$out/dump_join.pt PWR_JOIN_ONCE=0 python tools/race_hunt.py $var > $out/hunt_join.txt 2>&1; head -2 $out/hunt_join.txt | tail -1
