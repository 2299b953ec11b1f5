#!/bin/bash
# Copies the outputs of tools/profile_round6.sh (gpurun_out/prof_r06) into profiles/ under the round's names: one profile set per measured state.
P=gpurun_out/prof_r06
set -e
tail -1 $P/bench_default.txt > profiles/r06_bench_default.json
tail -1 $P/bench_driver_args_env_only.txt > profiles/r06_bench_driver_args_env_only.json
tail -1 $P/bench_under_rocprof.txt > profiles/r06_bench_under_rocprof.json
cp $P/bench_default_kernel_stats.csv profiles/r06_bench_default_kernel_stats.csv
cp $P/pmc_summary.json profiles/r06_pmc_summary.json
cp $P/learner_rooflines.txt profiles/r06_learner_rooflines.json
cp $P/learner_kernels_kernel_stats.csv profiles/r06_learner_kernels_kernel_stats.csv
cp $P/train_step_kernel_stats.csv profiles/r06_train_step_kernel_stats.csv
cp $P/rollout_pass_kernel_stats.csv profiles/r06_rollout_pass_kernel_stats.csv
tail -1 $P/bench_forward_search_config5.txt > profiles/r06_bench_forward_search_config5.json
cp $P/update_step_ops.txt profiles/r06_update_step_ops.txt
grep -v "amdgpu.ids" $P/k_step_timeline.txt > profiles/r06_k_step_timeline.txt
grep -v "amdgpu.ids" $P/k_step_type_split.txt > profiles/r06_k_step_type_split.txt
grep -v "amdgpu.ids" $P/lr_finish_profile.txt > profiles/r06_lr_finish_profile.txt
[ -f $P/k_step_sq_counters.json ] && cp $P/k_step_sq_counters.json profiles/r06_k_step_sq_counters.json
{ echo "# tools/soak_deferred.py 4 8000 1 (the fused-sampling loop, the default) and 2 8000 0 (sampler + k_step): ALL 65 536 games against the oracle after every leg"; grep -v "amdgpu.ids" $P/soak_fused.txt; grep -v "amdgpu.ids" $P/soak_sampler_loop.txt; } > profiles/r06_soak_all_games.txt
{ echo "# tools/pass_experiments.py on the final library: the default (fused-sampling) loop and CATAN_DEFERRED_FUSED=0 (sampler + k_step, the round-5 default)"; cat $P/pass_default.txt $P/pass_sampler_loop.txt; } > profiles/r06_pass_default_vs_sampler_loop.txt
{ cat $P/status.txt; tail -3 $P/gpu_tests.txt; for f in $P/*.err; do if [ -s $f ]; then echo "---- $(basename $f)"; tail -5 $f | cut -c1-200; fi; done; } > profiles/r06_profile_session_status.txt
cp $P/gpu_tests.txt profiles/r06_gpu_tests_final_library.txt
tail -1 $P/pass_default.txt
