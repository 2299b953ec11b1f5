/*
 * catan_hip_tuning.h - scheduling knobs, counters and profilers of the env kernels in libcatan_hip.so.  Benchmarks, sweeps and
 * diagnostics only (bench.py, tools): results never depend on any of them, and nothing a reference-side binding needs is
 * declared here (that is catan_hip.h).
 */
#ifndef CATAN_HIP_TUNING_H
#define CATAN_HIP_TUNING_H

#include "catan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scheduling knob of k_step (results do not depend on it): games per wave, 64 / 32 / 16.  With fewer games per wave there are
 * 2 / 4 waves per SIMD at 65 536 games, so that one wave's record transfers overlap the other waves' dependent-instruction
 * chains.  Default DEFAULT_STEP_WAVE_GAMES (csrc/catan_abi.hip), or the environment variable CATAN_STEP_WAVE_GAMES at creation. */
int catan_set_step_wave_games(catan_env_t* env, int32_t games);
/* tier-1 longest-road search budget (iterations) before a request is handed to tier 2: lock-step / deferred mode */
int catan_set_lr_budgets(catan_env_t* env, int32_t lockstep, int32_t deferred);
/* cumulative slow-path counters since creation (synchronises the stream): out3 = { longest-road requests handled by tier 1
 * (k_lr_finish), requests handed on to tier 2 (k_lr_heavy), k_lr_finish launches } - bench.py derives the bytes a launch moves */
int catan_slow_path_counts(catan_env_t* env, catan_stream_t stream, uint64_t* out3);
/* tier-2 longest-road search: iterations per bulk-synchronous round (work is re-shared between rounds): lock-step / deferred */
int catan_set_lr_rounds(catan_env_t* env, int32_t lockstep, int32_t deferred);

/* the rollout loops with a hipEvent around every kernel launch (recorded on the stream the kernel runs on); window <= 0: the
 * lock-step loop (step_idx0 as in catan_random_rollout), window > 0: the deferred loop (step_idx0 ignored).  kernel_ms is a HOST
 * float[5] receiving the summed milliseconds of k_sample_random (which also sorts the games by action type), k_step,
 * k_lr_finish, k_lr_heavy, k_reset_list / k_install_list (bench.py roofline). */
int catan_random_rollout_timed(catan_env_t* env, uint32_t step_idx0, int64_t steps, int32_t window, catan_stream_t stream, float* kernel_ms);
/* diagnostics of the lock-step step: finished games whose speculatively dealt successor was missing and which
 * were re-dealt on the critical path instead; expected 0 (k_step's may-end filter is a superset of the games a step can end) */
int64_t catan_missed_speculation_count(catan_env_t* env, catan_stream_t stream);

/* diagnostics: copies `bytes` (multiple of 16) device to device with one kernel (k_calib_copy) - a launch with exactly
 * known HBM traffic, used to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (profiles/README.md) */
int catan_calib_copy(void* dst, const void* src, int64_t bytes, catan_stream_t stream);

/* phase profile (diagnostics): enable (zeroes the counters) / read.  out16 (62 words) = 8 sums then 8 maxima, 4 tier-1
 * counters, then per action type (14) the sum / count / maximum of k_step's validate+apply time.
 * Slots 0, 1, 2, 6, 7: k_step phases per wave in 100 MHz wall-clock ticks (stage-in, validate+apply, request push,
 * done/reward+masks, write-back); slots 3, 4, 5: k_reset_list (philox draws, re-deals, serial shuffle ticks);
 * tools/phase_profile.py prints them. */
int catan_profile_enable(catan_env_t* env, int on);
int catan_profile_read(catan_env_t* env, uint64_t* out16);
/* catan_profile_enable(env, 2): contention-free variant for k_step - every wave stores its own phase durations of the LAST
 * launch; out: HOST uint32 [max(ceil(n/256)*16 + 17, 7128)][8] (k_step's waves first - slots 0,1,2,6,7 as above in 100 MHz ticks, slot 5 = sort bin + 1: bins
 * 0..12 = action types, 13..16 = play_dev with card 1..4; 17 = one partial wave per bin of the sort; rows 4128.. / 6128..: k_lr_finish's requests / k_lr_heavy's
 * workgroups of the last launch at 65 536 games) */
int catan_profile_read_waves(catan_env_t* env, uint32_t* out);
/* catan_profile_enable(env, 3): as 2, but slot 2 = the wave's START time (low 32 bits of the 100 MHz wall clock at entry; the
 * phases 0, 1, 6, 7 follow it back to back) and slot 3 = where the wave ran (HW_REG_HW_ID bits 0..27 | HW_REG_XCC_ID << 28):
 * the launch's timeline - dispatch ramp, waves that share a SIMD, the tail (tools/step_timeline.py). */

/* hipRuntimeGetVersion() of the HIP runtime this process runs on, or -1 (policy._Branches keys its hipGraph work-around on it: DESIGN.md 4.6) */
int32_t catan_hip_runtime_version(void);

/* Algorithmic HBM bytes of one fused env step per stepped game, from the static_assert-ed layout constants of csrc/catan_state.h
 * (action row in, hot record in, masks + reward + done out, the ideal write-back): bench.py's roofline numerator. */
int32_t catan_step_algorithmic_bytes(void);
/* ... and of the fused-sampling step (the game's action and decision counter out of its side row, the next action and the new masks back into it) */
int32_t catan_step_fused_algorithmic_bytes(void);
/* Which form of the deferred rollout loop catan_random_rollout_deferred runs (results are identical, game for game):
 *   1 (default since round 6)  fused sampling: k_step draws each completed game's next action into its side row and appends the game to the next
 *                pass's lists (per bin CATAN_FUSED_SUBS sub-lists with a counter each, so that the launch's waves do not queue up on one address);
 *                ONE kernel per pass on the main stream, tier 1 forks once per two passes
 *   0            a sampling + sorting kernel in front of every k_step (rounds 1-5) */
int catan_set_deferred_fused(catan_env_t* env, int32_t on);
int32_t catan_deferred_fused(const catan_env_t* env);

/* Environment switches read at catan_create (A/B diagnostics of the schedules; results never depend on them; defaults are the measured best,
 * DESIGN.md 4.0 / profiles/r05_s5_pass_experiments.txt):
 *   CATAN_STEP_BIN_ORDER=0        k_step's bins over the launch's waves in index order instead of longest-lasting first
 *   CATAN_LR_MID_BUDGET=b         budget of the middle tier of a deferred window (default 256; 0: every tier-2 request straight to k_lr_heavy)
 *   CATAN_LR_MID_HEAVY_GRID=g     workgroups of k_lr_heavy behind the middle tier (default 32)
 *   CATAN_T1_GROUP=1              one tier-1 launch per pass of the library's own deferred loop instead of one per two passes
 *   CATAN_T1_DEPTH=2              ... then with two rotating tier-1 slots instead of three
 *   CATAN_LR_SPLIT=0 | 2          tier 1 as search + lane-per-game completion never / in every schedule (default: where a launch has two passes)
 *   CATAN_LR_GRID=g               workgroups of k_lr_finish (default 4 096 inside a lock-step step, 3 072 in the deferred schedules)
 *   CATAN_STEP_WAVES_PER_BLOCK=4  four-wave k_step workgroups;  CATAN_STEP_WAVE_GAMES, CATAN_DEFERRED_FUSED: as the setters above
 *   CATAN_FUSED_SUBS=s            sub-lists per sort bin in the fused-sampling loop: 1, 2, 4, 8 (default) or 16
 *   CATAN_T1_DELAY_US=k           the fused-sampling loop: tier 1 staggered k microseconds behind the group's last k_step (default 4; 0: not staggered)
 *   CATAN_DEBUG_FUSED_CLOSE_UNORDERED=1, CATAN_DEBUG_STEP_DELAY_US=k   the fused loop's window close as it was ordered until round 6 / the closing pass's
 *                                 k_step k microseconds late: reproduce the round-5 parity failures at will (tools/fused_close_race.py; these two DO change results) */

#ifdef __cplusplus
}
#endif

#endif /* CATAN_HIP_TUNING_H */
