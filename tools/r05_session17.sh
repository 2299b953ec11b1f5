#!/bin/bash
# glue of the learner step: packed masks, one concatenation for log-probs / entropies, catan_recurrent_given - parity tests, then the step time
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status17.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status17.txt; }
run gpu_tests_17 timeout 1200 python -m pytest tests/test_gpu_ppo_pipeline.py tests/test_gpu_policy_fixture.py tests/test_gpu_obs_ppo.py tests/test_gpu_reference_api.py -q -m gpu -x
cd /tmp
run step_ops_17 timeout 600 python $R/tools/profile_update_step.py
cat $O/status17.txt; tail -5 $O/gpu_tests_17.txt; head -12 $O/step_ops_17.txt
