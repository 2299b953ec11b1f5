#!/bin/bash
# the sampler polls the tier-1 flag in memory instead of an event wait in the main stream's queue: parity, then the pass
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status26.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status26.txt; }
run gpu_tests_26 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py tests/test_gpu_abi_errors.py -q -m gpu -x
cd /tmp
run p26_flag timeout 200 python $R/tools/pass_experiments.py
run p26_event env CATAN_LR_FLAG=0 timeout 200 python $R/tools/pass_experiments.py
run p26_flag_d3 env CATAN_T1_DEPTH=3 timeout 200 python $R/tools/pass_experiments.py
run p26_flag2 timeout 200 python $R/tools/pass_experiments.py
run p26_event2 env CATAN_LR_FLAG=0 timeout 200 python $R/tools/pass_experiments.py
cat $O/status26.txt; tail -3 $O/gpu_tests_26.txt; for f in p26_flag p26_event p26_flag_d3 p26_flag2 p26_event2; do tail -1 $O/$f.txt | cut -c1-330; done
