#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status19.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status19.txt; }
run gpu_tests_19 timeout 600 python -m pytest tests/test_gpu_ppo_pipeline.py -q -m gpu -x -k "gathered_parts or compact or rollout_and_update"
cd /tmp
run ab_19 env SWITCHES=wgrad_big timeout 600 python $R/tools/ab_step_switches.py 12
cat $O/status19.txt; tail -3 $O/gpu_tests_19.txt; cat $O/ab_19.txt
