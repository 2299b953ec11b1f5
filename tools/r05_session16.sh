#!/bin/bash
# full GPU suite on the tree with the recomputing backward (off by default), the ABI checks and the log-prob gap test
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status16.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status16.txt; }
run gpu_tests_16 timeout 1700 python -m pytest tests -q -m gpu -s -k "not config3"
cat $O/status16.txt; grep -n "logp_learner" $O/gpu_tests_16.txt | head -3; tail -6 $O/gpu_tests_16.txt
