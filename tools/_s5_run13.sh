# session 5 / run 13: threads per sampler workgroup; instruction-cache counters of k_step
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run13.txt; : > $O
for cfg in "" "CATAN_SAMPLER_BLOCK=128" "CATAN_SAMPLER_BLOCK=512" "CATAN_SAMPLER_BLOCK=1024" ""; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/pass_experiments.py 2>&1 | tail -1 >> $O
done
bash tools/pmc_k_step_icache.sh >> $O 2>&1
cat $O
