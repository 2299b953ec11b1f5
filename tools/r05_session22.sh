#!/bin/bash
# tier-1 DFS stacks of 24 levels (5.5 KB of LDS per k_lr_finish workgroup instead of 10.4): parity, then the pass
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status22.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status22.txt; }
run gpu_tests_22 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py tests/test_gpu_collector.py -q -m gpu -x
cd /tmp
run p22_d2 timeout 200 python $R/tools/pass_experiments.py
run p22_d3 env CATAN_T1_DEPTH=3 timeout 200 python $R/tools/pass_experiments.py
cat $O/status22.txt; tail -3 $O/gpu_tests_22.txt; for f in p22_d2 p22_d3; do tail -1 $O/$f.txt | cut -c1-300; done
