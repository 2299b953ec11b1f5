#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status4.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status4.txt; }
cd /tmp
run pass_wpb4 timeout 200 python $R/tools/pass_experiments.py
run pass_wpb1 env CATAN_STEP_WAVES_PER_BLOCK=1 timeout 200 python $R/tools/pass_experiments.py
run pass_wpb4_fused env CATAN_DEFERRED_FUSED=1 timeout 200 python $R/tools/pass_experiments.py
run pass_wpb1_fused env CATAN_DEFERRED_FUSED=1 CATAN_STEP_WAVES_PER_BLOCK=1 timeout 200 python $R/tools/pass_experiments.py
run pass_skip1 env CATAN_DEBUG_SKIP=1 timeout 200 python $R/tools/pass_experiments.py
run pass_skip2 env CATAN_DEBUG_SKIP=2 timeout 200 python $R/tools/pass_experiments.py
run step_timeline_wpb4 timeout 300 python $R/tools/step_timeline.py
cd $R
run gpu_tests_4 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py tests/test_gpu_collector.py -x -q -m gpu
cat $O/status4.txt; for f in pass_wpb4 pass_wpb1 pass_wpb4_fused pass_wpb1_fused pass_skip1 pass_skip2; do tail -1 $O/$f.txt | cut -c1-400; done; head -8 $O/step_timeline_wpb4.txt; tail -3 $O/gpu_tests_4.txt
