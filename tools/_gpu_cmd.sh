mkdir -p gpurun_out/r2g
python -m pytest tests -q -m gpu -x > gpurun_out/r2g/tests.log 2>&1; tail -6 gpurun_out/r2g/tests.log
for v in 0 1 2; do PWR_DEC_FWD128=$v PWR_DEC_BWD128=$v python tools/bench_decoder.py 2>/dev/null | sed "s/^/variant $v: /" >> gpurun_out/r2g/decoder.jsonl; done; cat gpurun_out/r2g/decoder.jsonl
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --accuracy-steps 0"
for cfg in "PWR_WGRAD3_SLOTS=256 PWR_WGRAD_TR_SLOTS=512" "PWR_WGRAD3_SLOTS=128 PWR_WGRAD_TR_SLOTS=512" "PWR_WGRAD3_SLOTS=256 PWR_WGRAD_TR_SLOTS=256" "PWR_WGRAD3_SLOTS=128 PWR_WGRAD_TR_SLOTS=256" "PWR_WGRAD3_SLOTS=128 PWR_WGRAD_TR_SLOTS=128" "PWR_WGRAD3_SLOTS=64 PWR_WGRAD_TR_SLOTS=128" "PWR_WGRAD3_SLOTS=256 PWR_WGRAD_TR_SLOTS=512"; do echo "$cfg: $(env $cfg $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step, infer", round(d["infer_frames_per_s"]))')" >> gpurun_out/r2g/splits.txt; done; cat gpurun_out/r2g/splits.txt
