# session 5 / run 12: after the fix of the split path's stale mark (pend.len uninitialised; the cut case left the word alone): the shard test x4, the suite
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run12.txt; : > $O
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_env_parity.py -m gpu -q -k "shard_invariance or reproducible" 2>&1 | tail -1 >> $O; done
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s5/gpu_tests_run12.txt 2>&1; echo "gpu tests rc=$?" >> $O; tail -3 gpurun_out/s5/gpu_tests_run12.txt >> $O
echo "==" >> $O
timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
cat $O
