#!/bin/bash
# Round 5, first measured state: the whole -m gpu suite, the default bench line, the k_step timeline and SQ counters.
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status.txt; }
run gpu_tests_1 timeout 1500 python -m pytest tests -x -q -m gpu
cd /tmp
run bench_1 timeout 700 python $R/bench.py
run step_timeline timeout 300 python $R/tools/step_timeline.py
run step_timeline_fused env FUSED=1 timeout 300 python $R/tools/step_timeline.py
run pmc_k_step bash $R/tools/pmc_k_step.sh
cat $O/status.txt; tail -4 $O/gpu_tests_1.txt; head -c 400 $O/bench_1.txt; echo; cat $O/step_timeline.txt
