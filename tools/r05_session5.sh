#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status5.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status5.txt; }
cd /tmp
run pass_base timeout 200 python $R/tools/pass_experiments.py
run pass_extstop env CATAN_EXT_STOP_EVENT=1 timeout 200 python $R/tools/pass_experiments.py
run trace_extstop env CATAN_EXT_STOP_EVENT=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_extstop -o t -- python $R/tools/pmc_workload.py
run pass_gaps_extstop python $R/tools/pass_gaps.py $O/trace_extstop 64
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
run gpu_tests_5 env CATAN_EXT_STOP_EVENT=1 timeout 900 python -m pytest tests/test_gpu_env_parity.py -x -q -m gpu
cat $O/status5.txt; for f in pass_base pass_extstop; do tail -1 $O/$f.txt | cut -c1-400; tail -2 $O/$f.err; done; cat $O/pass_gaps_extstop.txt; tail -3 $O/gpu_tests_5.txt
