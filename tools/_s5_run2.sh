# session 5 / run 2: tier-1 split (search + lane-per-game completion), k_step with reserved AGPRs, two-wave workgroups
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run2.txt; : > $O
echo "== parity, CATAN_LR_SPLIT=1" >> $O
CATAN_LR_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_golden.py tests/test_gpu_abi_errors.py -m gpu -x -q 2>&1 | tail -5 >> $O
for cfg in "" "CATAN_LR_SPLIT=1" "CATAN_STEP_AGPR=96" "CATAN_STEP_AGPR=160" "CATAN_STEP_WAVES_PER_BLOCK=2" "CATAN_LR_SPLIT=1 CATAN_STEP_AGPR=96" "CATAN_LR_SPLIT=1 CATAN_T1_DEPTH=3" "CATAN_LR_SPLIT=1 CATAN_STEP_AGPR=96 CATAN_T1_DEPTH=3" ""; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
