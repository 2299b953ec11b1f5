#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status27.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status27.txt; }
cd /tmp
run p27_event env CATAN_LR_FLAG=0 timeout 200 python $R/tools/pass_experiments.py
run p27_flag timeout 200 python $R/tools/pass_experiments.py
run p27_flag_b8 env LR_BUDGET=8 timeout 200 python $R/tools/pass_experiments.py
run p27_flag_b6 env LR_BUDGET=6 timeout 200 python $R/tools/pass_experiments.py
run p27_flag_d3 env CATAN_T1_DEPTH=3 timeout 200 python $R/tools/pass_experiments.py
run p27_flag_d3_b8 env CATAN_T1_DEPTH=3 LR_BUDGET=8 timeout 200 python $R/tools/pass_experiments.py
run p27_event_b8 env CATAN_LR_FLAG=0 LR_BUDGET=8 timeout 200 python $R/tools/pass_experiments.py
cat $O/status27.txt; for f in p27_event p27_flag p27_flag_b8 p27_flag_b6 p27_flag_d3 p27_flag_d3_b8 p27_event_b8; do tail -1 $O/$f.txt | cut -c1-330; tail -1 $O/$f.err | cut -c1-160; done
