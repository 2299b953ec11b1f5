#!/bin/bash
# Round-5 profiles on the MI355X box (run through gpurun from the repo root).  Outputs land in gpurun_out/prof_r05; every step's exit code goes
# to status.txt and a non-empty stderr tail to <step>.err (a step that dies leaves a trace: VERDICT r4, weak #2).  --pmc passes never share a
# run with a trace domain.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r05
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
run() { name=$1; shift; ( "$@" ) > $OUT/$name.txt 2> $OUT/$name.err.full; rc=$?; echo "$name rc=$rc" >> $OUT/status.txt; grep -v "amdgpu.ids" $OUT/$name.err.full | tail -30 > $OUT/$name.err; rm -f $OUT/$name.err.full; [ -s $OUT/$name.err ] || rm -f $OUT/$name.err; return 0; }
run gpu_tests timeout 2400 python -m pytest tests -q -m gpu
cd /tmp
run bench_default timeout 900 python $REPO/bench.py
run bench_driver_args_env_only timeout 300 python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --ppo-steps 0 --no-live-pmc
run bench_under_rocprof timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-live-pmc --ppo-steps 0
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_default_kernel_stats.csv \;
run pmc_fetch timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $REPO/tools/pmc_workload.py
run pmc_write timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $REPO/tools/pmc_workload.py
run pmc_summarise python $REPO/tools/pmc_summarise.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write
run learner_rooflines timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/learner -o learner -- python $REPO/tools/learner_rooflines_workload.py
find $OUT/learner -name "*kernel_stats.csv" -exec cp {} $OUT/learner_kernels_kernel_stats.csv \;
run train_step timeout 900 env STEPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python $REPO/tools/pmc_policy_workload.py
find $OUT/train -name "*kernel_stats.csv" -exec cp {} $OUT/train_step_kernel_stats.csv \;
run rollout_pass timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roll -o roll -- python $REPO/tools/profile_rollout_pass.py
find $OUT/roll -name "*kernel_stats.csv" -exec cp {} $OUT/rollout_pass_kernel_stats.csv \;
run bench_forward_search_config5 timeout 600 python $REPO/tools/bench_forward_search.py --decisions 1
run rollout_schedules timeout 900 python $REPO/tools/rollout_schedules.py
run update_step_ops timeout 600 python $REPO/tools/profile_update_step.py
run ab_step_switches timeout 900 env SWITCHES=te_fused_bwd,wgrad_big,recompute_h,grouped_wgrad python $REPO/tools/ab_step_switches.py 12
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*agent_info.csv" -delete
rm -rf $OUT/stats $OUT/learner $OUT/train $OUT/roll
cat $OUT/status.txt; tail -3 $OUT/gpu_tests.txt; head -c 300 $OUT/bench_default.txt; echo; ls $OUT/*.err 2>/dev/null
