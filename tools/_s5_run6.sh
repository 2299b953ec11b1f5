# session 5 / run 6: new defaults (bin order, middle tier 256 / 32, tier-1 depth 3) + one call site per topology function in compute_masks
# + batched cold-list accesses in play_dev: the whole GPU suite, then the pass A/B against the old schedule switches
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run6.txt; : > $O
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s5/gpu_tests_run6.txt 2>&1; echo "gpu tests rc=$?" >> $O; tail -3 gpurun_out/s5/gpu_tests_run6.txt >> $O
for cfg in "" "CATAN_T1_DEPTH=2" "CATAN_STEP_BIN_ORDER=0 CATAN_LR_MID_BUDGET=0 CATAN_T1_DEPTH=2" ""; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
echo "== timeline" >> $O
timeout 300 python tools/step_timeline.py 2>&1 | tail -34 >> $O
cat $O
