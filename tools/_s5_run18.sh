# session 5 / run 18: the sampler reserves its block's list ranges between the type draw and the rest of the action (sample_random = head + tail)
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run18.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py tests/test_gpu_collector.py -m gpu -q 2>&1 | tail -2 >> $O
for i in 1 2 3; do timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O; done
cat $O
