#!/bin/bash
# Round-5 baseline of the tree as restored: GPU tests, bench line, kernel stats, PMC traffic, the six rollout schedules, step anatomy.
# Every step's stdout / stderr / exit code is kept (profile steps that die leave a trace: VERDICT r4 weak #2c).
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status11.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status11.txt; }
run gpu_tests_11 timeout 1700 python -m pytest tests -x -q -m gpu
cd /tmp
run bench_11 timeout 900 python $R/bench.py
run bench_rocprof timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --no-cpu-baseline --no-live-pmc --ppo-steps 0
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_default_kernel_stats.csv \;
run rollout_schedules timeout 900 python $R/tools/rollout_schedules.py
run update_step_ops timeout 600 python $R/tools/profile_update_step.py
run ab_wgrad_big env SWITCHES=wgrad_big timeout 600 python $R/tools/ab_step_switches.py 12
run learner_rooflines timeout 600 python $R/tools/learner_rooflines_workload.py
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; rm -rf $O/stats
cat $O/status11.txt; tail -4 $O/gpu_tests_11.txt; head -c 400 $O/bench_11.txt; echo; cat $O/rollout_schedules.txt | cut -c1-200; tail -3 $O/rollout_schedules.err; head -12 $O/update_step_ops.txt; cat $O/ab_wgrad_big.txt
