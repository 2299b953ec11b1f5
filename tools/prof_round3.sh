mkdir -p gpurun_out/p1
python tools/profile_modules.py > gpurun_out/p1/modules.log 2>&1
python tools/profile_rollout_pass.py > gpurun_out/p1/rollout_pass.log 2>&1
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1/roll -o roll -- python $R/tools/profile_rollout_pass.py > $R/gpurun_out/p1/roll.log 2>&1
STEPS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1/train -o train -- python $R/tools/pmc_policy_workload.py > $R/gpurun_out/p1/train.log 2>&1
cd $R; find gpurun_out/p1 -name "*kernel_trace.csv" -delete; find gpurun_out/p1 -name "*.csv" -size +4M -delete
cat gpurun_out/p1/modules.log gpurun_out/p1/rollout_pass.log
