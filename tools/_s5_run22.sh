# session 5 / run 22: shorter windows with a smaller middle-tier budget
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run22.txt; : > $O
for cfg in "WINDOW=32" "WINDOW=24 CATAN_LR_MID_BUDGET=128" "WINDOW=24 CATAN_LR_MID_BUDGET=64" "WINDOW=20 CATAN_LR_MID_BUDGET=64" "WINDOW=16 CATAN_LR_MID_BUDGET=48" "WINDOW=32 CATAN_LR_MID_BUDGET=128" "WINDOW=32"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 | cut -c1-260 >> $O
done
cat $O
