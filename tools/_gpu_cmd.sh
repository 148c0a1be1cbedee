mkdir -p gpurun_out/r2j
python -m pytest tests/test_ops_gpu.py -q -x -k "persistent or conv_forward or dgrad or epilogue" > gpurun_out/r2j/tests.log 2>&1; tail -5 gpurun_out/r2j/tests.log
PWR_PATCH_PERSIST=0 python tools/bench_kernels.py all 20 > gpurun_out/r2j/iso_off.jsonl 2>/dev/null; PWR_PATCH_PERSIST=1 python tools/bench_kernels.py all 20 > gpurun_out/r2j/iso_on.jsonl 2>/dev/null; cat gpurun_out/r2j/iso_off.jsonl gpurun_out/r2j/iso_on.jsonl | cut -c1-200
B="python bench.py --steps 150 --warmup 20 --no-cpu-baseline --accuracy-steps 0"
for cfg in "PWR_PATCH_PERSIST=1" "PWR_PATCH_PERSIST=0" "PWR_PATCH_PERSIST=1" "PWR_PATCH_PERSIST=0"; do echo "$cfg: $(env $cfg $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step, infer", round(d["infer_frames_per_s"]), "roofline", round(d["roofline"]["frac"],3), round(d["roofline"]["us_per_launch"],1))')" >> gpurun_out/r2j/ab.txt; done; cat gpurun_out/r2j/ab.txt
