#!/bin/bash
# the tier-1 stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask): does k_step get its CUs back?
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status21.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status21.txt; }
cd /tmp
run p21_base timeout 200 python $R/tools/pass_experiments.py
for k in 128 64 32 16; do
  run p21_cus$k env CATAN_LR_CUS=$k timeout 200 python $R/tools/pass_experiments.py
  run p21_cus${k}_d3 env CATAN_LR_CUS=$k CATAN_T1_DEPTH=3 timeout 200 python $R/tools/pass_experiments.py
done
run p21_cus32_block env CATAN_LR_CUS=32 CATAN_LR_CUS_PATTERN=block timeout 200 python $R/tools/pass_experiments.py
cat $O/status21.txt; for f in p21_base p21_cus128 p21_cus128_d3 p21_cus64 p21_cus64_d3 p21_cus32 p21_cus32_d3 p21_cus16 p21_cus16_d3 p21_cus32_block; do tail -1 $O/$f.txt | cut -c1-300; tail -1 $O/$f.err | cut -c1-200; done
