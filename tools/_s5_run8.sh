# session 5 / run 8: the env parity file three times on the defaults (the middle tier out of the fused loop again), then tier 1 forked once per two passes
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run8.txt; : > $O
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -m gpu -q 2>&1 | tail -2 >> $O; done
echo "== parity, CATAN_T1_GROUP=2" >> $O
CATAN_T1_GROUP=2 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -m gpu -q 2>&1 | tail -3 >> $O
for cfg in "" "CATAN_T1_GROUP=2" "" "CATAN_T1_GROUP=2"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
