export TMPDIR=/tmp; R=$(pwd); mkdir -p $R/gpurun_out/p2; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p2/heads -o heads -- python $R/tools/profile_heads.py > $R/gpurun_out/p2/heads.log 2>&1
cd $R; find gpurun_out/p2 -name "*kernel_trace.csv" -delete; tail -5 gpurun_out/p2/heads.log
