# session 5 / run 9: workgroups of k_lr_finish under the new defaults
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run9.txt; : > $O
for cfg in "" "CATAN_LR_GRID=3072" "CATAN_LR_GRID=2048" "CATAN_LR_GRID=1536" "CATAN_LR_GRID=1024" ""; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
