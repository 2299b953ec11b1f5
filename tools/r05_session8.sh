#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status8.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status8.txt; }
cd /tmp
run p8_base timeout 200 python $R/tools/pass_experiments.py
run p8_prio env CATAN_STEP_PRIO=1 timeout 200 python $R/tools/pass_experiments.py
run p8_grid2048 env CATAN_LR_GRID=2048 timeout 200 python $R/tools/pass_experiments.py
run p8_grid1024 env CATAN_LR_GRID=1024 timeout 200 python $R/tools/pass_experiments.py
run p8_prio_stop_grid2048 env CATAN_STEP_PRIO=1 CATAN_EXT_STOP_EVENT=1 CATAN_LR_GRID=2048 timeout 200 python $R/tools/pass_experiments.py
run p8_w8 env WINDOW=8 timeout 200 python $R/tools/pass_experiments.py
run p8_w16 env WINDOW=16 timeout 200 python $R/tools/pass_experiments.py
cat $O/status8.txt; for f in p8_base p8_prio p8_grid2048 p8_grid1024 p8_prio_stop_grid2048 p8_w8 p8_w16; do tail -1 $O/$f.txt | cut -c1-330; done
