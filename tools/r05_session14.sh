#!/bin/bash
# the recomputing tile-encoder backward: parity test against the sub-layer kernels, then forward / backward times at a minibatch's board count
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status14.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status14.txt; }
run te_fused_small env BOARDS=1037 timeout 300 python $R/tools/bench_te_fused_bwd.py
run gpu_tests_14 timeout 900 python -m pytest tests/test_gpu_ppo_pipeline.py -q -m gpu -x -k "tile_encoder"
run te_fused_bench timeout 300 python $R/tools/bench_te_fused_bwd.py
cat $O/status14.txt; cat $O/te_fused_small.txt; tail -3 $O/te_fused_small.err; tail -15 $O/gpu_tests_14.txt; cat $O/te_fused_bench.txt; tail -3 $O/te_fused_bench.err
