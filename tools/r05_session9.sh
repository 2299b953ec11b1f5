#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status9.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status9.txt; }
run gpu_tests_9 timeout 1500 python -m pytest tests -x -q -m gpu
cd /tmp
run stress_determinism timeout 900 python $R/tools/stress_determinism.py 30
run bench_9 timeout 700 python $R/bench.py
cat $O/status9.txt; tail -4 $O/gpu_tests_9.txt; cat $O/stress_determinism.txt; tail -3 $O/stress_determinism.err; head -c 300 $O/bench_9.txt
