/*
 * catan_hip.h - C ABI of libcatan_hip.so: the MI355X-native batched Catan environment.
 *
 * Drop-in boundary.  The upstream reference (henrycharlesworth/settlers_of_catan_RL) has no FFI; its de-facto
 * boundary is the Python surface of EnvWrapper (env/wrapper.py).  Each entry point below names the reference
 * interface it replaces (file:line).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C, no exceptions, no torch types: every buffer is a caller-owned DEVICE pointer (e.g. torch tensor
 *    .data_ptr()); calls are asynchronous on the given HIP stream (pass NULL for the default stream);
 *    one handle per GPU / per thread.
 *  - return 0 on success, negative CATAN_E* otherwise; catan_last_error() gives the message (thread local).
 *  - n = number of games of the handle.  Batched layouts are head-major / field-major so that the lane-per-game
 *    kernels read and write them coalesced:
 *        actions  int32 [18][n]   the 12-head composite action of env/wrapper.py:114-166 flattened
 *                                 ([0]type [1]corner [2]edge(72=dummy) [3]tile [4]dev card [5]accept0/reject1
 *                                  [6]relative player [7..10]give seq [11..14]receive seq [15]res A [16]res B [17]discard)
 *        reward   float [4][n]    index = PlayerId-1 (White, Blue, Orange, Red; game/enums.py:8-12)
 *        done     uint8 [n]
 *        masks    float [n][325]  the 12 arrays of get_action_masks (env/wrapper.py:172-185) concatenated
 *        blob     int32 [736][cnt] canonical full state (settlers_of_catan_rl_amd/spec.py STATE_FIELDS)
 *  - RNG: per-game Philox4x32-10 streams keyed by (seed, env_id0 + game index): results do not depend on how
 *    games are sharded over GPUs (DESIGN.md "RNG").
 */
#ifndef CATAN_HIP_H
#define CATAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct catan_env catan_env_t;
typedef void* catan_stream_t; /* hipStream_t */

enum { CATAN_OK = 0, CATAN_EINVAL = -1, CATAN_ENOMEM = -2, CATAN_EHIP = -3, CATAN_ENODEV = -4 };

/* EnvWrapper.__init__ keyword arguments, env/wrapper.py:12-13 */
typedef struct {
    int32_t max_proposed_trades_per_turn; /* 4; negative = None (unlimited) */
    float win_reward;                     /* 500 */
    int32_t dense_reward;                 /* 0 */
    float reward_annealing_factor;        /* 1.0 (env.reward_annealing_factor, RL/ppo/game_manager.py:164-166) */
    int32_t validate_actions;             /* 1: an action whose mask bit is clear is rejected and counted */
    int32_t auto_reset;                   /* 1: a finished game is reset inside catan_step (game_manager.py:112-113) */
} catan_cfg_t;

void catan_cfg_default(catan_cfg_t* cfg);

/* constants of the flat layouts */
int32_t catan_state_words(void);      /* 736 */
int32_t catan_mask_words(void);       /* 325 */
int32_t catan_action_words(void);     /* 18  */
int32_t catan_obs_floats(void);       /* 1787 */
int32_t catan_state_bytes_per_game(void); /* packed HBM bytes per game */

/* EnvWrapper() x n + reset(): env/wrapper.py:24,30-34.  Games are reset and their masks computed. */
int catan_create(catan_env_t** out, int device, int64_t n_envs, uint64_t seed, uint64_t env_id0, const catan_cfg_t* cfg);
void catan_destroy(catan_env_t* env);
const char* catan_last_error(void);
int64_t catan_num_envs(const catan_env_t* env);

/* EnvWrapper.reset(): env/wrapper.py:30-34.  reset_mask (uint8[n], device) selects games; NULL = all. */
int catan_reset(catan_env_t* env, const uint8_t* reset_mask, catan_stream_t stream);

/* EnvWrapper.step(action): env/wrapper.py:36-50 (translate + apply + done/reward); then, per cfg, auto-reset of
 * finished games; then the next legal-action masks are refreshed (kept packed inside the handle). */
int catan_step(catan_env_t* env, const int32_t* actions, float* reward, uint8_t* done, catan_stream_t stream);

/* EnvWrapper.get_action_masks(): env/wrapper.py:168-290, batched float32 [n][325]. */
int catan_masks(catan_env_t* env, float* out_masks, catan_stream_t stream);
/* the same masks as 325-bit strings: uint32 [11][pitch] (bit i of the flat mask = word i>>5, bit i&31) */
int catan_masks_packed(catan_env_t* env, const uint32_t** out_ptr, int64_t* out_pitch);

/* deciding player (discarder > trade target > players_go): env/wrapper.py:53-58, RL/ppo/game_manager.py:152-159.
 * out: int32 [n], PlayerId 1..4 */
int catan_deciding_seat(catan_env_t* env, int32_t* out, catan_stream_t stream);

/* uniform-random legal policy used by bench config 2 (DESIGN.md "random policy"); writes int32 [18][n] */
int catan_sample_random_actions(catan_env_t* env, uint32_t step_idx, int32_t* actions, catan_stream_t stream);

/* EnvWrapper.save_state()/restore_state(): env/wrapper.py:711-721 -> game/game.py:1013-1205.
 * env_idx: int64 [cnt] device game ids, NULL = games 0..cnt-1. */
int catan_state_export(catan_env_t* env, int32_t* blob, const int64_t* env_idx, int64_t cnt, catan_stream_t stream);
int catan_state_import(catan_env_t* env, const int32_t* blob, const int64_t* env_idx, int64_t cnt, catan_stream_t stream);

/* env.reward_annealing_factor = f (RL/ppo/game_manager.py:164-166) */
int catan_set_reward_annealing(catan_env_t* env, float factor);
/* number of rejected (mask-illegal) actions since creation; synchronises the stream.  The reference raises
 * RuntimeError on the first one (env/wrapper.py:38-41). */
int64_t catan_invalid_action_count(catan_env_t* env, catan_stream_t stream);

/* bench helper: `steps` x (sample_random_actions -> step) on one stream with the handle's own scratch buffers.
 * total_done (int64 device scalar, may be NULL) accumulates finished games. */
int catan_random_rollout(catan_env_t* env, uint32_t step_idx0, int64_t steps, catan_stream_t stream);

/* the same loop with a hipEvent pair around every kernel launch (recorded on `stream`); kernel_ms is a HOST
 * float[4] receiving the summed milliseconds of k_sample_random, k_step, k_lr_heavy, k_step_finish (bench.py roofline). */
int catan_random_rollout_timed(catan_env_t* env, uint32_t step_idx0, int64_t steps, catan_stream_t stream, float* kernel_ms);

/* k_step phase profile (diagnostics): enable (zeroes the counters) / read.  out16 = 8 sums over waves then 8
 * per-wave maxima, in 100 MHz wall-clock ticks, for the phases stage-in, validate+apply, tier-1 longest road,
 * holder logic (+cut), done/reward, reset, masks, write-back. */
int catan_profile_enable(catan_env_t* env, int on);
int catan_profile_read(catan_env_t* env, uint64_t* out16);

#ifdef __cplusplus
}
#endif
#endif
