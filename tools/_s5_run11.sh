# session 5 / run 11: the whole GPU suite on the new defaults (grouped tier 1, 3 072 tier-1 workgroups and the search / completion split in deferred schedules)
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run11.txt; : > $O
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s5/gpu_tests_run11.txt 2>&1; echo "gpu tests rc=$?" >> $O; tail -4 gpurun_out/s5/gpu_tests_run11.txt >> $O
for cfg in "" "CATAN_LR_SPLIT=0"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
