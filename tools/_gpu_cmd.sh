mkdir -p gpurun_out/r2i
B="python bench.py --steps 150 --warmup 20 --no-cpu-baseline --accuracy-steps 0"
for cfg in "PWR_NORM_BWD_PAR=0 PWR_NORM_PAR=1" "PWR_NORM_BWD_PAR=2 PWR_NORM_PAR=1" "PWR_NORM_BWD_PAR=0 PWR_NORM_PAR=0" "PWR_NORM_BWD_PAR=1 PWR_NORM_PAR=1" "PWR_NORM_BWD_PAR=0 PWR_NORM_PAR=1" "PWR_NORM_BWD_PAR=2 PWR_NORM_PAR=1"; do echo "$cfg: $(env $cfg $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step, infer", round(d["infer_frames_per_s"]))')" >> gpurun_out/r2i/ab.txt; done; cat gpurun_out/r2i/ab.txt
