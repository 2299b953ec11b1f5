#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status20.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status20.txt; }
run gpu_tests_20 timeout 1700 python -m pytest tests -x -q -m gpu
run smoke_20 timeout 300 python -c "import __graft_entry__ as g; g.smoke()"
cat $O/status20.txt; tail -3 $O/gpu_tests_20.txt; tail -2 $O/smoke_20.txt
