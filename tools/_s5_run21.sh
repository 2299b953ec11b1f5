# session 5 / run 21: the deferred window under the final schedule (WINDOW = passes per window)
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run21.txt; : > $O
for cfg in "WINDOW=32" "WINDOW=16" "WINDOW=24" "WINDOW=48" "WINDOW=64" "WINDOW=32"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 | cut -c1-260 >> $O
done
cat $O
