#!/bin/bash
# Round-4 profiles (run on the GPU box from the repo root): per-kernel statistics of the train step (three streams, and everything on
# one stream = the serial kernel time), of `bench.py --roofline-only` and of the isolated hot kernels, HBM traffic of the dominant conv
# and of the decoder from separate PMC passes, MFMA utilisation.
#   bash tools/profile_r4.sh  ->  gpurun_out/r4p/*  (summaries -> gpurun_out/r4p/summary by tools/profile_summary.py, then copied to profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o step -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline --accuracy-steps 0 > $O/step_bench.json 2> $O/step.err
PWR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python3 $R/bench.py --debug-lib --steps 25 --warmup 5 --no-cpu-baseline --accuracy-steps 0 > $O/serial_bench.json 2> $O/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -o roof -- python3 $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/roof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso -o iso -- python3 $R/tools/bench_kernels.py all 20 80 > $O/iso_bench.jsonl 2> $O/iso.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso24 -o iso24 -- python3 $R/tools/bench_kernels.py wgrad 20 24 > $O/iso24_bench.jsonl 2> $O/iso24.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -o dec -- python3 $R/tools/bench_decoder.py > $O/dec_bench.jsonl 2> $O/dec.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/bench_kernels.py all 3 80 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/bench_kernels.py all 3 80 > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_dec -o f -- python3 $R/tools/bench_decoder.py > /dev/null 2> $O/pmc_fetch_dec.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_dec -o w -- python3 $R/tools/bench_decoder.py > /dev/null 2> $O/pmc_write_dec.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- python3 $R/tools/bench_kernels.py all 3 80 > /dev/null 2> $O/pmc_mfma.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma24 -o m -- python3 $R/tools/bench_kernels.py wgrad 3 24 > /dev/null 2> $O/pmc_mfma24.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch24 -o f -- python3 $R/tools/bench_kernels.py wgrad 3 24 > /dev/null 2> $O/pmc_fetch24.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write24 -o w -- python3 $R/tools/bench_kernels.py wgrad 3 24 > /dev/null 2> $O/pmc_write24.err
cd $R
python3 tools/profile_summary.py $O $O/summary r4 > $O/summary.log 2>&1; tail -60 $O/summary.log
# one step's kernel timeline (queue, start, duration, workgroups, kernel) out of the step trace; host issue time; the side-stream layers alone
python3 tools/step_timeline.py $O/step $O/summary/r4_step_timeline.tsv 5 > $O/summary/r4_step_timeline.txt 2>&1; cat $O/summary/r4_step_timeline.txt
python3 tools/host_issue.py > $O/summary/r4_host_issue.json 2>/dev/null; cat $O/summary/r4_host_issue.json
python3 tools/bench_side.py all 20 2>/dev/null | grep layer > $O/summary/r4_side_layers.jsonl; cat $O/summary/r4_side_layers.jsonl
# keep the merge-back small: the raw traces are large
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
du -sh $O
# the GPU suite, the default bench line (accuracy block and CPU baseline included) and the smoke test of the same build, on the same lease
python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $O/summary/r4_pytest_gpu.txt; cat $O/summary/r4_pytest_gpu.txt
python bench.py 2>/dev/null > $O/summary/r4_bench.json; cut -c1-330 $O/summary/r4_bench.json
python tools/lease_check.py 2>&1 | grep "product library" > $O/summary/r4_lease_check.txt; cat $O/summary/r4_lease_check.txt
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
