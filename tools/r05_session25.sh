#!/bin/bash
# what does ONE more event record per pass on the main stream cost? (results stay correct)
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status25.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status25.txt; }
cd /tmp
for k in 0 1 2 4; do run p25_extra$k env CATAN_DEBUG_EXTRA_EVENTS=$k timeout 200 python $R/tools/pass_experiments.py; done
for k in 0 1 2 4; do tail -1 $O/p25_extra$k.txt | cut -c1-260; done
