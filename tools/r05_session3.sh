#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status3.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status3.txt; }
run gpu_tests_3 timeout 900 python -m pytest tests/test_gpu_ppo_pipeline.py tests/test_optim.py tests/test_gpu_policy_fixture.py -x -q -m gpu
cd /tmp
run step_timeline timeout 300 python $R/tools/step_timeline.py
run step_timeline_fused env FUSED=1 timeout 300 python $R/tools/step_timeline.py
run trace_passes timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_passes -o t -- python $R/tools/pmc_workload.py
run pass_gaps python $R/tools/pass_gaps.py $O/trace_passes 64
run trace_passes_fused env CATAN_DEFERRED_FUSED=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_passes_fused -o t -- python $R/tools/pmc_workload.py
run pass_gaps_fused python $R/tools/pass_gaps.py $O/trace_passes_fused 64
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
run ab_recompute_h env SWITCHES=recompute_h timeout 600 python $R/tools/ab_step_switches.py 16
cat $O/status3.txt; tail -4 $O/gpu_tests_3.txt; cat $O/step_timeline.txt | head -12; cat $O/pass_gaps.txt; cat $O/pass_gaps_fused.txt; cat $O/ab_recompute_h.txt; tail -3 $O/ab_recompute_h.err
