# session 5: the round's final profile set on the library as committed: tools/profile_round5.sh, then k_step's SQ counters, wave timeline and per-type split
bash tools/profile_round5.sh > gpurun_out/prof_r05_session.log 2>&1
bash tools/pmc_k_step.sh > gpurun_out/prof_r05/k_step_sq.log 2>&1
cp gpurun_out/r05/k_step_sq_counters.json gpurun_out/prof_r05/ 2>/dev/null; rm -rf gpurun_out/r05/k_step_pmc
bash tools/pmc_k_step_icache.sh > gpurun_out/prof_r05/k_step_icache.log 2>&1
cp gpurun_out/r05/k_step_icache_counters.json gpurun_out/prof_r05/ 2>/dev/null; rm -rf gpurun_out/r05/k_step_icache
cd $GRAFT_REPO_ROOT
timeout 300 python tools/step_timeline.py > gpurun_out/prof_r05/k_step_timeline.txt 2>&1
timeout 300 python tools/step_type_split.py > gpurun_out/prof_r05/k_step_type_split.txt 2>&1
timeout 300 python tools/pass_experiments.py > gpurun_out/prof_r05/pass_default.txt 2>&1
find gpurun_out -name '*counter_collection.csv' -size +2M -delete; du -sh gpurun_out; tail -20 gpurun_out/prof_r05_session.log
