# session 5 / run 7: the fused-sampling loop (CATAN_DEFERRED_FUSED=1) with the round's new schedule pieces
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run7.txt; : > $O
echo "== parity, fused + middle tier" >> $O
CATAN_DEFERRED_FUSED=1 CATAN_LR_MID_FUSED=1 timeout 900 python -m pytest tests/test_gpu_env_parity.py -m gpu -x -q 2>&1 | tail -3 >> $O
for cfg in "" "CATAN_DEFERRED_FUSED=1" "CATAN_DEFERRED_FUSED=1 CATAN_LR_MID_FUSED=1" "CATAN_DEFERRED_FUSED=1 CATAN_LR_MID_FUSED=1 CATAN_STEP_BIN_ORDER=0"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
