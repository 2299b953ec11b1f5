#!/bin/bash
# tier-1 depth 2 vs 3 on ONE box, alternating
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status23.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status23.txt; }
cd /tmp
for i in 1 2 3; do
  run p23_d2_$i timeout 200 python $R/tools/pass_experiments.py
  run p23_d3_$i env CATAN_T1_DEPTH=3 timeout 200 python $R/tools/pass_experiments.py
done
for i in 1 2 3; do tail -1 $O/p23_d2_$i.txt | cut -c1-200; tail -1 $O/p23_d3_$i.txt | cut -c1-200; done
