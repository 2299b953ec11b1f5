# session 5 / run 14: branch-free topology helpers (catan_topology.inc regenerated): parity, pass, per-type split
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run14.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -m gpu -q 2>&1 | tail -2 >> $O
for i in 1 2; do timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O; done
timeout 300 python tools/step_type_split.py 2>&1 | grep -v amdgpu >> $O
cat $O
