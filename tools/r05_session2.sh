#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status2.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status2.txt; }
run gpu_tests_2 timeout 900 python -m pytest tests/test_gpu_ppo_pipeline.py tests/test_optim.py tests/test_gpu_policy_fixture.py -x -q -m gpu
cd /tmp
run step_timeline timeout 300 python $R/tools/step_timeline.py
run step_timeline_fused env FUSED=1 timeout 300 python $R/tools/step_timeline.py
run ab_recompute_h env SWITCHES=recompute_h timeout 600 python $R/tools/ab_step_switches.py 16
cat $O/status2.txt; tail -4 $O/gpu_tests_2.txt; cat $O/step_timeline.txt; tail -3 $O/step_timeline.err; cat $O/step_timeline_fused.txt | head -12; cat $O/ab_recompute_h.txt; tail -3 $O/ab_recompute_h.err
