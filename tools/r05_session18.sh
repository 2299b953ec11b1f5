#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status18.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status18.txt; }
run gpu_tests_18 timeout 1200 python -m pytest tests/test_gpu_ppo_pipeline.py tests/test_gpu_policy_fixture.py tests/test_gpu_obs_ppo.py tests/test_gpu_reference_api.py -q -m gpu -x
cd /tmp
run step_ops_18 env STACKS=1 timeout 600 python $R/tools/profile_update_step.py
cat $O/status18.txt; tail -5 $O/gpu_tests_18.txt; head -3 $O/step_ops_18.txt; grep -n "small launches" -A45 $O/step_ops_18.txt | cut -c1-200
