#!/bin/bash
# Round-6 profiles on the MI355X box (run through gpurun from the repo root): ONE set for the library as committed.  Outputs land in
# gpurun_out/prof_r06; every step's exit code goes to status.txt and a non-empty stderr tail to <step>.err.  --pmc passes never share a run
# with a trace domain.  tools/copy_profile_set.sh copies what is judged into profiles/r06_*.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r06
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
run() { name=$1; shift; ( "$@" ) > $OUT/$name.txt 2> $OUT/$name.err.full; rc=$?; echo "$name rc=$rc" >> $OUT/status.txt; grep -v "amdgpu.ids" $OUT/$name.err.full | tail -30 > $OUT/$name.err; rm -f $OUT/$name.err.full; [ -s $OUT/$name.err ] || rm -f $OUT/$name.err; return 0; }
run gpu_tests timeout 2400 python -m pytest tests -q -m gpu
run smoke timeout 600 python -c "import __graft_entry__ as g; g.smoke()"
cd /tmp
run bench_default timeout 1200 python $REPO/bench.py
run bench_driver_args_env_only timeout 300 python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --ppo-steps 0 --no-live-pmc
run bench_under_rocprof timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-live-pmc --ppo-steps 0
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_default_kernel_stats.csv \;
run pmc_fetch timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $REPO/tools/pmc_workload.py
run pmc_write timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $REPO/tools/pmc_workload.py
run pmc_summarise python $REPO/tools/pmc_summarise.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write
run pass_default timeout 300 python $REPO/tools/pass_experiments.py
run pass_sampler_loop env CATAN_DEFERRED_FUSED=0 timeout 300 python $REPO/tools/pass_experiments.py
run k_step_timeline timeout 600 python $REPO/tools/step_timeline.py
run k_step_type_split timeout 600 python $REPO/tools/step_type_split.py
run lr_finish_profile timeout 300 python $REPO/tools/lr_finish_profile.py
run soak_fused timeout 900 python $REPO/tools/soak_deferred.py 4 8000 1
run soak_sampler_loop timeout 900 python $REPO/tools/soak_deferred.py 2 8000 0
run fused_close_race_unordered env CATAN_DEBUG_FUSED_CLOSE_UNORDERED=1 CATAN_DEBUG_STEP_DELAY_US=60 timeout 300 python $REPO/tools/fused_close_race.py
run fused_close_race_ordered env CATAN_DEBUG_STEP_DELAY_US=60 timeout 300 python $REPO/tools/fused_close_race.py
run learner_rooflines timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/learner -o learner -- python $REPO/tools/learner_rooflines_workload.py
find $OUT/learner -name "*kernel_stats.csv" -exec cp {} $OUT/learner_kernels_kernel_stats.csv \;
run train_step timeout 900 env STEPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python $REPO/tools/pmc_policy_workload.py
find $OUT/train -name "*kernel_stats.csv" -exec cp {} $OUT/train_step_kernel_stats.csv \;
run rollout_pass timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roll -o roll -- python $REPO/tools/profile_rollout_pass.py
find $OUT/roll -name "*kernel_stats.csv" -exec cp {} $OUT/rollout_pass_kernel_stats.csv \;
run bench_forward_search_config5 timeout 600 python $REPO/tools/bench_forward_search.py --decisions 1
run update_step_ops timeout 600 python $REPO/tools/profile_update_step.py
export R=$REPO
sed -e 's#gpurun_out/r05/k_step_pmc#gpurun_out/prof_r06/k_step_pmc#' $REPO/tools/pmc_k_step.sh > /tmp/pmc_k_step_r06.sh
run k_step_sq timeout 1200 bash /tmp/pmc_k_step_r06.sh
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*agent_info.csv" -delete
rm -rf $OUT/stats $OUT/learner $OUT/train $OUT/roll
cat $OUT/status.txt; tail -3 $OUT/gpu_tests.txt; head -c 400 $OUT/bench_default.txt; echo; ls $OUT/*.err 2>/dev/null
