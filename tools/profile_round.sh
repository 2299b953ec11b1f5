#!/bin/bash
# Round profiles on the MI355X box (run through gpurun from the repo root): kernel-trace stats of the default bench
# command, then the HBM counters in their own passes (--pmc is never combined with a trace domain; FETCH_SIZE and
# WRITE_SIZE do not fit one pass), then the kernel-trace of the fused learner kernels.  Outputs land in
# gpurun_out/prof_r02; the summaries are copied to profiles/ afterwards.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-live-pmc --ppo-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1
python $REPO/tools/pmc_summarise.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo -o ppo -- python $REPO/tools/ppo_kernels_workload.py > $OUT/ppo.log 2>&1
find $OUT/ppo -name "*kernel_stats.csv" -exec cp {} $OUT/ppo_kernels_kernel_stats.csv \;
# keep the merge-back small: the per-dispatch traces are large
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls -la $OUT; tail -3 $OUT/pmc_summary.log; head -c 400 $OUT/bench_default.json; grep -E "k_gae|k_ppo_loss|k_adv" $OUT/ppo_kernels_kernel_stats.csv | cut -c1-200
