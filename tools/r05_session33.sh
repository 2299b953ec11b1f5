#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status33.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status33.txt; }
run gpu_tests_33 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -q -m gpu -x
cd /tmp
run p33_a timeout 200 python $R/tools/pass_experiments.py
run p33_b timeout 200 python $R/tools/pass_experiments.py
cat $O/status33.txt; tail -2 $O/gpu_tests_33.txt; for f in p33_a p33_b; do tail -1 $O/$f.txt | cut -c1-300; done
