# session 5 / run 19: 32 / 16 games per k_step wave (2 / 4 waves per SIMD) under the round's final schedule
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run19.txt; : > $O
for cfg in "" "CATAN_STEP_WAVE_GAMES=32" "CATAN_STEP_WAVE_GAMES=16" "CATAN_STEP_WAVE_GAMES=32 CATAN_LR_GRID=2048" ""; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
cat $O
