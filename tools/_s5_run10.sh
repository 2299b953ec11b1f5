# session 5 / run 10: 3 072 tier-1 workgroups in deferred schedules (built in); tier 1 split into search + lane-per-game completion again, now that a
# tier-1 launch has two passes to finish
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run10.txt; : > $O
for cfg in "" "CATAN_LR_SPLIT=1" "CATAN_LR_SPLIT=1 CATAN_LR_GRID=2048" "" "CATAN_LR_SPLIT=1"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
