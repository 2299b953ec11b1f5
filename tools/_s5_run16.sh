# session 5 / run 17: + the accepted trade in closed form on the twenty hand bytes (respond)
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run17.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -m gpu -q 2>&1 | tail -2 >> $O
for i in 1 2 3; do timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O; done
timeout 300 python tools/step_type_split.py 2>&1 | grep -E "^roll|roll's" >> $O
cat $O
