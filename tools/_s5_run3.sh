# session 5 / run 3: k_step's bins laid over the waves longest-lasting first (CATAN_STEP_BIN_ORDER=1)
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run3.txt; : > $O
echo "== parity, CATAN_STEP_BIN_ORDER=1" >> $O
CATAN_STEP_BIN_ORDER=1 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3 >> $O
for cfg in "" "CATAN_STEP_BIN_ORDER=1" "CATAN_STEP_BIN_ORDER=1 CATAN_T1_DEPTH=3" "" "CATAN_STEP_BIN_ORDER=1"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
echo "== timeline, default order" >> $O
timeout 300 python tools/step_timeline.py 2>&1 | tail -32 >> $O
echo "== timeline, CATAN_STEP_BIN_ORDER=1" >> $O
CATAN_STEP_BIN_ORDER=1 timeout 300 python tools/step_timeline.py 2>&1 | tail -32 >> $O
cat $O
