#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status32.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status32.txt; }
run gpu_tests_32 timeout 900 python -m pytest tests/test_gpu_ppo_pipeline.py tests/test_gpu_policy_fixture.py -q -m gpu -x
cd /tmp
run ab_32 env SWITCHES=te_ends timeout 600 python $R/tools/ab_step_switches.py 12
cat $O/status32.txt; tail -4 $O/gpu_tests_32.txt; cat $O/ab_32.txt
