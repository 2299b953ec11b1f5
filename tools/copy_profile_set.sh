#!/bin/bash
# Copies the outputs of tools/_s5_final.sh (gpurun_out/prof_r05) into profiles/ under the round's names: one profile set per measured state.
P=gpurun_out/prof_r05
set -e
tail -1 $P/bench_default.txt > profiles/r05_bench_default.json
tail -1 $P/bench_driver_args_env_only.txt > profiles/r05_bench_driver_args_env_only.json
tail -1 $P/bench_under_rocprof.txt > profiles/r05_bench_under_rocprof.json
cp $P/bench_default_kernel_stats.csv profiles/r05_bench_default_kernel_stats.csv
cp $P/pmc_summary.json profiles/r05_pmc_summary.json
cp $P/learner_rooflines.txt profiles/r05_learner_rooflines.json
cp $P/learner_kernels_kernel_stats.csv profiles/r05_learner_kernels_kernel_stats.csv
cp $P/train_step_kernel_stats.csv profiles/r05_train_step_kernel_stats.csv
cp $P/rollout_pass_kernel_stats.csv profiles/r05_rollout_pass_kernel_stats.csv
tail -1 $P/bench_forward_search_config5.txt > profiles/r05_bench_forward_search_config5.json
cp $P/rollout_schedules.txt profiles/r05_rollout_schedules.txt
cp $P/update_step_ops.txt profiles/r05_update_step_ops.txt
cp $P/ab_step_switches.txt profiles/r05_ab_step_switches.txt
cp $P/k_step_sq_counters.json profiles/r05_k_step_sq_counters.json
cp $P/k_step_icache_counters.json profiles/r05_k_step_icache_counters.json
grep -v "amdgpu.ids" $P/k_step_timeline.txt > profiles/r05_k_step_timeline.txt
grep -v "amdgpu.ids" $P/k_step_type_split.txt > profiles/r05_k_step_type_split.txt
{ cat $P/status.txt; tail -3 $P/gpu_tests.txt; for f in $P/*.err; do if [ -s $f ]; then echo "---- $(basename $f)"; tail -5 $f | cut -c1-200; fi; done; } > profiles/r05_profile_session_status.txt
cp $P/gpu_tests.txt profiles/r05_gpu_tests_final_library.txt
tail -1 $P/pass_default.txt
