# session 5 / run 20: games per k_step wave x tier-1 workgroups, two rounds
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run20.txt; : > $O
for rep in 1 2; do
for cfg in "" "CATAN_STEP_WAVE_GAMES=32" "CATAN_STEP_WAVE_GAMES=32 CATAN_LR_GRID=2048" "CATAN_STEP_WAVE_GAMES=16 CATAN_LR_GRID=2048" "CATAN_LR_GRID=2048"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 | cut -c1-260 >> $O
done; done
cat $O
