#!/bin/bash
# tier-1 depth 3 with smaller k_lr_finish grids (its duration is off the critical cycle at depth 3: fewer waves = less LDS taken from k_step)
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status15.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status15.txt; }
cd /tmp
for g in 4096 2048 1024 512 256; do
  run p15_d3_g$g env CATAN_T1_DEPTH=3 CATAN_LR_GRID=$g timeout 200 python $R/tools/pass_experiments.py
done
run p15_d2_g1024 env CATAN_LR_GRID=1024 timeout 200 python $R/tools/pass_experiments.py
cat $O/status15.txt; for g in 4096 2048 1024 512 256; do tail -1 $O/p15_d3_g$g.txt | cut -c1-330; done; tail -1 $O/p15_d2_g1024.txt | cut -c1-330
