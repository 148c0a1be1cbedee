#!/bin/bash
# Round-6 profiles (run on the GPU box from the repo root).  FIRST the known-answer digests: a box that computes other bytes says which kernel
# family before anything is measured on it (and is then diagnosed: tools/kat.py --diagnose = repeat the differing cases, lease_check --bisect,
# race_hunt).  Then: per-kernel statistics of TRAIN STEPS ONLY (tools/train_steps.py: calls / 41 = launches per step), of `bench.py
# --roofline-only`, of the isolated hot kernels and of the decoder; HBM traffic and MFMA utilisation of the dominant conv from separate PMC
# passes; chain time per network part (debug build's scope events); the GPU suite, the default bench line and the smoke test on the same lease.
#   bash tools/profile_r6.sh  ->  gpurun_out/r6p/*  (summaries -> gpurun_out/r6p/summary by tools/profile_summary.py; copy those to profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6p; mkdir -p $O/summary
cd $R
python tools/kat.py > $O/summary/r6_kat.json 2>$O/kat.err || python tools/kat.py --diagnose > $O/summary/r6_kat_diagnose.txt 2>&1
cat $O/summary/r6_kat.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trainsteps -o t -- python3 $R/tools/train_steps.py 40 > $O/trainsteps.json 2> $O/trainsteps.err
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
cd $R
python3 tools/profile_summary.py $O $O/summary r6 > $O/summary.log 2>&1; tail -40 $O/summary.log
python3 tools/step_breakdown.py 30 > $O/summary/r6_step_breakdown.json 2>/dev/null; head -c 600 $O/summary/r6_step_breakdown.json
python3 tools/stamp_wstat.py 32 2>/dev/null | tail -10 > $O/summary/r6_wstat_phases.txt; python3 tools/stamp_wstat.py 32 stats 2>/dev/null | tail -10 >> $O/summary/r6_wstat_phases.txt; cat $O/summary/r6_wstat_phases.txt
ELIM_STEPS=60 python3 tools/step_elimination.py > $O/summary/r6_step_elimination.json 2>/dev/null; head -c 900 $O/summary/r6_step_elimination.json
python3 tools/host_issue.py > $O/summary/r6_host_issue.json 2>/dev/null
python3 tools/bench_side.py all 20 2>/dev/null | grep layer > $O/summary/r6_side_layers.jsonl
# keep the merge-back small: the raw traces are large
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
du -sh $O
python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $O/summary/r6_pytest_gpu.txt; tail -5 $O/summary/r6_pytest_gpu.txt
python bench.py 2>/dev/null > $O/summary/r6_bench.json; cut -c1-330 $O/summary/r6_bench.json
python tools/lease_check.py 2>&1 | grep "product library" > $O/summary/r6_lease_check.txt; cat $O/summary/r6_lease_check.txt
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
