#!/bin/bash
# tier-1 depth 3 vs 2 in the library's own deferred loop; the GPU tests that did not run in session 11 (it stopped at test_optim)
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status12.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status12.txt; }
cd /tmp
run p12_d3 timeout 200 python $R/tools/pass_experiments.py
run p12_d2 env CATAN_T1_DEPTH=2 timeout 200 python $R/tools/pass_experiments.py
run p12_d3_w64 env WINDOW=64 timeout 200 python $R/tools/pass_experiments.py
run p12_d3_w16 env WINDOW=16 timeout 200 python $R/tools/pass_experiments.py
cd $R
run gpu_tests_12a timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py tests/test_gpu_abi_errors.py tests/test_optim.py -q -m gpu
run gpu_tests_12b timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_env_parity.py --deselect tests/test_gpu_golden.py
cat $O/status12.txt; for f in p12_d3 p12_d2 p12_d3_w64 p12_d3_w16; do tail -1 $O/$f.txt | cut -c1-330; done; tail -4 $O/gpu_tests_12a.txt; tail -4 $O/gpu_tests_12b.txt
