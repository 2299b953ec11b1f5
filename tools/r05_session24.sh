#!/bin/bash
# what do the two queue packets per pass cost?  (timing only: without the hand-overs the games' results are wrong)
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status24.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status24.txt; }
cd /tmp
run p24_base timeout 200 python $R/tools/pass_experiments.py
run p24_nowait env CATAN_DEBUG_NOEVENTS=1 timeout 200 python $R/tools/pass_experiments.py
run p24_norecord env CATAN_DEBUG_NOEVENTS=2 timeout 200 python $R/tools/pass_experiments.py
run p24_neither env CATAN_DEBUG_NOEVENTS=3 timeout 200 python $R/tools/pass_experiments.py
cat $O/status24.txt; for f in p24_base p24_nowait p24_norecord p24_neither; do tail -1 $O/$f.txt | cut -c1-300; tail -2 $O/$f.err | cut -c1-200; done
