#!/bin/bash
# where the small launches of a minibatch step come from (stack attribution); the optimiser tests
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status13.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status13.txt; }
run gpu_tests_13 timeout 600 python -m pytest tests/test_optim.py tests/test_gpu_env_parity.py -q -m gpu
cd /tmp
run update_step_stacks env STACKS=1 timeout 900 python $R/tools/profile_update_step.py
cat $O/status13.txt; tail -3 $O/gpu_tests_13.txt; grep -n "small launches" -A70 $O/update_step_stacks.txt | cut -c1-220
