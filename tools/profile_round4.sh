#!/bin/bash
# Round-4 profiles on the MI355X box (run through gpurun from the repo root).  Outputs land in gpurun_out/prof_r04; the summaries
# are copied to profiles/r04_* afterwards.  --pmc passes never share a run with a trace domain.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r04
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --ppo-steps 0 --no-live-pmc > $OUT/bench_driver_args_env_only.json 2> $OUT/bench_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-live-pmc --ppo-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_default_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1
python $REPO/tools/pmc_summarise.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_learner_fetch -o pmc -- python $REPO/tools/pmc_learner_workload.py > $OUT/pmc_learner_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_learner_write -o pmc -- python $REPO/tools/pmc_learner_workload.py > $OUT/pmc_learner_write.log 2>&1
PMC_TAIL=4 PMC_KEEP_TEMPLATE_ARGS=1 python $REPO/tools/pmc_summarise.py $OUT/pmc_learner_summary.json $OUT/pmc_learner_fetch $OUT/pmc_learner_write > $OUT/pmc_learner_summary.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/learner -o learner -- python $REPO/tools/learner_rooflines_workload.py > $OUT/learner_rooflines.json 2> $OUT/learner.err
find $OUT/learner -name "*kernel_stats.csv" -exec cp {} $OUT/learner_kernels_kernel_stats.csv \;
STEPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python $REPO/tools/pmc_policy_workload.py > $OUT/train.log 2>&1
find $OUT/train -name "*kernel_stats.csv" -exec cp {} $OUT/train_step_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roll -o roll -- python $REPO/tools/profile_rollout_pass.py > $OUT/rollout_pass.log 2>&1
find $OUT/roll -name "*kernel_stats.csv" -exec cp {} $OUT/rollout_pass_kernel_stats.csv \;
python $REPO/tools/bench_forward_search.py --decisions 1 > $OUT/bench_forward_search_config5.json 2> $OUT/fs.err
python $REPO/tools/rollout_schedules.py > $OUT/rollout_schedules.txt 2> $OUT/rollout_schedules.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/steptrace -o t -- python $REPO/tools/trace_update_steps_workload.py > $OUT/steptrace.log 2>&1
python $REPO/tools/trace_update_steps_summary.py $OUT/steptrace > $OUT/update_steps_trace_summary.txt 2>&1
python $REPO/tools/ab_step_switches.py 12 > $OUT/ab_step_switches.txt 2> $OUT/ab_step_switches.err
python $REPO/tools/profile_update_step.py > $OUT/update_step_ops.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*agent_info.csv" -delete
ls -la $OUT; tail -2 $OUT/pmc_summary.log; head -c 300 $OUT/bench_default.json; echo; tail -c 600 $OUT/bench_forward_search_config5.json; tail -3 $OUT/rollout_pass.log
