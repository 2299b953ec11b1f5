# session 5 / run 5: the middle tier (one-wave searches with a large budget over a window's tier-2 requests, k_lr_heavy for the rest)
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run5.txt; : > $O
echo "== parity, CATAN_LR_MID_BUDGET=128 CATAN_LR_MID_HEAVY_GRID=32 CATAN_STEP_BIN_ORDER=1" >> $O
CATAN_LR_MID_BUDGET=128 CATAN_LR_MID_HEAVY_GRID=32 CATAN_STEP_BIN_ORDER=1 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3 >> $O
for cfg in "CATAN_STEP_BIN_ORDER=1" "CATAN_STEP_BIN_ORDER=1 CATAN_LR_MID_BUDGET=64" "CATAN_STEP_BIN_ORDER=1 CATAN_LR_MID_BUDGET=128" "CATAN_STEP_BIN_ORDER=1 CATAN_LR_MID_BUDGET=128 CATAN_LR_MID_HEAVY_GRID=32" "CATAN_STEP_BIN_ORDER=1 CATAN_LR_MID_BUDGET=256 CATAN_LR_MID_HEAVY_GRID=32" "CATAN_STEP_BIN_ORDER=1 CATAN_LR_MID_BUDGET=512 CATAN_LR_MID_HEAVY_GRID=16" "CATAN_STEP_BIN_ORDER=1 CATAN_LR_MID_BUDGET=128 CATAN_LR_MID_HEAVY_GRID=32 CATAN_T1_DEPTH=3" "CATAN_STEP_BIN_ORDER=1"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
