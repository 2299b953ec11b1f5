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
 *  - n = number of games of the handle.  Batched layouts are game-major (one contiguous record per game: the step kernel
 *    gathers games sorted by action type, so everything belonging to a game sits in as few cache lines as possible):
 *        actions  int32 [n][18]   the 12-head composite action of env/wrapper.py:114-166 flattened
 *                                 ([0]type [1]corner [2]edge(72=dummy) [3]tile [4]dev card [5]accept0/reject1
 *                                  [6]relative player [7..10]give seq [11..14]receive seq [15]res A [16]res B [17]discard)
 *        reward   float [n][4]    index = PlayerId-1 (White, Blue, Orange, Red; game/enums.py:8-12)
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
    int32_t dense_reward;                 /* 0 */
    int32_t validate_actions;             /* 1: an action whose mask bit is clear is rejected and counted */
    int32_t auto_reset;                   /* 1: a finished game is reset inside catan_step (game_manager.py:112-113) */
    double win_reward;                    /* 500 */
    double reward_annealing_factor;       /* 1.0 (env.reward_annealing_factor, RL/ppo/game_manager.py:164-166).  Both are doubles
                                           * because the reference shapes rewards in Python floats (env/wrapper.py:95-110) and
                                           * rounds once, when the rollout tensors are built (RL/ppo/process_batch.py:63) */
    int32_t max_actions_per_turn;         /* negative = None (np.inf, the default); otherwise once game.actions_this_turn EXCEEDS
                                           * it only EndTurn stays legal after the roll (env/wrapper.py:12-17,233-234) */
    int32_t reserved_;                    /* keeps sizeof a multiple of 8; set to 0 */
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
/* The hash of the sources (every file under csrc/, include/catan_hip.h) this binary was built from, baked in by the build
 * (settlers_of_catan_rl_amd/_lib.py: build_library / source_hash); the loader refuses a binary whose hash differs from
 * the sources beside it - the .so is kept out of git but shipped in-tree, and file times mean nothing after a checkout. */
const char* catan_build_hash(void);
int64_t catan_num_envs(const catan_env_t* env);

/* EnvWrapper.reset(): env/wrapper.py:30-34.  reset_mask (uint8[n], device) selects games; NULL = all. */
int catan_reset(catan_env_t* env, const uint8_t* reset_mask, catan_stream_t stream);

/* RNG contract (A) of SURVEY.md 8.4 for a SINGLE-GAME handle (n_envs == 1; CATAN_EINVAL otherwise): from this call on the game draws from the
 * reference's two process-global Mersenne Twisters instead of its Philox stream - np.random.shuffle / np.random.randint (game/components/board.py:72-84,
 * game/game.py:42,77,139-140) from an MT19937 seeded as np.random.seed(numpy_seed) does (init_genrand, masked rejection on the low bits),
 * random.choice of the steal (game/game.py:643) from an MT19937 seeded as random.seed(python_seed) does (init_by_array, the TOP bits of an output).
 * Both generators live in device memory and are advanced by the library's kernels; nothing is drawn on the host.  The game itself is not touched:
 * what the reference's `np.random.seed(s); random.seed(s); env = EnvWrapper(); env.reset()` draws is catan_reset_board_only (the Board() constructor,
 * board.py:47), catan_reset (Game.__init__ -> reset, game.py:40) and catan_reset again (EnvWrapper.reset, env/wrapper.py:30-34).
 * catan_mt19937_set_state installs a generator's full state instead (which: 0 = numpy's, 1 = `random`'s; key[624], pos in 0..624: the tuples
 * np.random.get_state() / random.getstate() return).  Lock-step entry points only: the deferred calls, the deferred rollouts and
 * catan_randomise_uncertainty return CATAN_EINVAL for such a handle (the oracle has no MT form of randomise_uncertainty to pin one against);
 * catan_state_import does not rewind the generators (the reference's restore_state does not either). */
int catan_seed_mt19937(catan_env_t* env, uint32_t numpy_seed, uint32_t python_seed, catan_stream_t stream);
int catan_mt19937_set_state(catan_env_t* env, int32_t which, const uint32_t* key624, int32_t pos, catan_stream_t stream);
/* Board.reset alone (game/components/board.py:67-100): takes its draws; the game must be reset afterwards (see above). */
int catan_reset_board_only(catan_env_t* env, catan_stream_t stream);

/* EnvWrapper.step(action): env/wrapper.py:36-50 (translate + apply + done/reward); then, per cfg, auto-reset of
 * (a game whose action type is negative is left untouched: explicit no-op, reward 0, done 0)
 * finished games; then the next legal-action masks are refreshed (kept packed inside the handle). */
int catan_step(catan_env_t* env, const int32_t* actions, float* reward, uint8_t* done, catan_stream_t stream);

/* EnvWrapper.step(action) for collectors that advance every env independently (RL/ppo/game_manager.py:78-113: each env of a
 * worker steps on its own; nothing there waits for the slowest env).  catan_step completes every game's step before it returns -
 * its duration is that of the slowest longest-road search or re-deal among n games.  catan_step_deferred applies the same
 * actions, but a game whose step needs that slow path (Game.update_longest_road after a road / settlement, game/game.py:864-919;
 * the reset of a finished game) completes it on side streams while the other games go on, and WAITS meanwhile:
 *     status[g] = CATAN_STEP_COMPLETE  reward[g][4] / done[g] are the result of the game's last applied action (for a game that
 *                                      waited: the action it was given when it started waiting), its state, masks and
 *                                      observation are current, and it takes actions[g] of the next call;
 *     status[g] = CATAN_STEP_WAITING   the step is still being completed: reward / done are 0, the game's state must not be
 *                                      read, and actions[g] of the next call is ignored (not counted as invalid).
 * Each game goes through exactly the states catan_step would take it through for the same sequence of applied actions; only
 * the interleaving between games differs (tests/test_gpu_env_parity.py).  Which games wait, and for how long, is fixed by the
 * schedule (`window` calls per slow-path window: a longest-road game returns after 2 calls, a game that needed the second
 * search tier or a reset after 1-2 windows), never by kernel timing: a sequence is reproducible.
 * A sequence of calls (same window, same stream) is closed by catan_step_flush, which completes every outstanding step:
 * status CATAN_STEP_COMPLETE + reward / done for the games that were waiting, CATAN_STEP_NONE (reward / done 0) for the others.
 * While a sequence is open, catan_step / reset / state_export / state_import / random_rollout* / randomise_uncertainty
 * return CATAN_EINVAL; catan_masks / obs / obs_rows / deciding_seat are valid for the games that are not waiting.
 * With catan_set_reward_f64_buffer the unrounded rewards of a game are written when its step completes (read them for
 * CATAN_STEP_COMPLETE games only).  status: uint8 [n]. */
enum { CATAN_STEP_COMPLETE = 0, CATAN_STEP_WAITING = 1, CATAN_STEP_NONE = 2 };
int catan_step_deferred(catan_env_t* env, const int32_t* actions, int32_t window, float* reward, uint8_t* done, uint8_t* status,
                        catan_stream_t stream);
int catan_step_flush(catan_env_t* env, float* reward, uint8_t* done, uint8_t* status, catan_stream_t stream);

/* EnvWrapper.get_action_masks(): env/wrapper.py:168-290, batched float32 [n][325]. */
int catan_masks(catan_env_t* env, float* out_masks, catan_stream_t stream);
/* the same masks as 325-bit strings: uint32 [n][pitch], pitch AS RETURNED in *out_pitch (currently 32 words = one 128-byte line per game: words 0..10
 * are the mask bits - bit i of the flat mask = word i>>5, bit i&31 - the rest is the library's own side data; never hard-code the pitch) */
int catan_masks_packed(catan_env_t* env, const uint32_t** out_ptr, int64_t* out_pitch);
/* the masks of catan_masks as packed rows of 11 words, uint32 [n][11] (what the rollout storage keeps per decision) */
int catan_masks_packed_copy(catan_env_t* env, uint32_t* out, catan_stream_t stream);
/* dst[t[r]][r][:] = src[r][:] for the rows r with sel[r] != 0: "append this game's observation / action / mask to its own
 * list" of RL/ppo/game_manager.py:102-133 for all games at once, the selected rows found on the device (no host read of how
 * many there are).  dst: [steps][rows][row_bytes] with `step_stride_bytes` between steps; t: int64 [rows] step index per row
 * (in range for the selected rows); sel: uint8 [rows]. */
int catan_masked_row_store(void* dst, const void* src, const int64_t* t, const uint8_t* sel, int64_t rows, int64_t row_bytes,
                           int64_t step_stride_bytes, catan_stream_t stream);
/* packed 325-bit mask rows (uint32 [rows][pitch_words], pitch_words >= 11; the rollout storage keeps 11 words per decision)
 * -> float32 [rows][325], the layout `action_masks` has in RL/ppo/process_batch.py:96-104 */
int catan_expand_masks(const uint32_t* packed, int64_t rows, int32_t pitch_words, float* out_masks, catan_stream_t stream);

/* The per-game bookkeeping of GamesAndPoliciesManager.gather_rollouts (RL/ppo/game_manager.py:69-140) for all games at once, two
 * launches per env iteration (settlers_of_catan_rl_amd/rollout.py describes the four per-game counters that stand for the
 * reference's Python lists).  pre: a_env int32 [n][18] = the actions for catan_step, with the no-op type for frozen games (n_obs =
 * T + 1), and live uint8 [n].  post (after catan_step and catan_deciding_seat): :91-136 - terminal masks, the reward sums over a
 * turn (doubles), the append of the active seat's action / log-prob / packed action mask, of rewards and terminal masks into the
 * (T(+2), n, .) rollout tensors, finished games, and which games append their NEXT observation where (sel, t_obs: the inputs of
 * catan_obs_rows).  counters4: int64 [4][n] = n_obs, n_msk, n_act, n_rew; flags4: uint8 [4][n] = done_since, pending_obs, live
 * (from pre), sel (out); reward64 may be NULL (then `reward` is summed).
 * With catan_step_deferred: waiting_before = the status array the previous call returned (all zero before the first), status =
 * the one this call returned; the action / log-prob / mask append of a decision happens in the iteration that passes the action
 * to the env, everything that needs the step's result in the iteration that delivers it.  Both NULL with catan_step. */
int catan_collector_pre(int64_t n, int32_t T, const int64_t* n_obs, const int64_t* actions, int32_t* a_env, uint8_t* live, catan_stream_t stream);
int catan_collector_post(int64_t n, int32_t T, int64_t* counters4, double* racc, uint8_t* flags4, float* term, int64_t* t_obs, const int64_t* active_pid,
                         const int32_t* deciding, const int32_t* n_deciding, const int64_t* actions, const float* logp, const int32_t* pmasks,
                         const float* reward, const double* reward64, const uint8_t* done, int64_t* st_actions, float* st_logp, int32_t* st_amasks,
                         float* st_rewards, float* st_masks, int64_t* n_complete, const uint8_t* waiting_before, const uint8_t* status,
                         catan_stream_t stream);

/* deciding player (discarder > trade target > players_go): env/wrapper.py:53-58, RL/ppo/game_manager.py:152-159.
 * out: int32 [n], PlayerId 1..4 */
int catan_deciding_seat(catan_env_t* env, int32_t* out, catan_stream_t stream);
/* the forward-search simulator's notion of whose turn it is (trade target > players_go; the discard phase is NOT
 * looked at): RL/forward_search_policy/worker.py:146-151.  out: int32 [n], PlayerId 1..4 */
int catan_players_turn_sim(catan_env_t* env, int32_t* out, catan_stream_t stream);

/* uniform-random legal policy used by bench config 2 (DESIGN.md "random policy"); writes int32 [n][18] */
int catan_sample_random_actions(catan_env_t* env, uint32_t step_idx, int32_t* actions, catan_stream_t stream);

/* EnvWrapper.save_state()/restore_state(): env/wrapper.py:711-721 -> game/game.py:1013-1205.
 * env_idx: int64 [cnt] device game ids, NULL = games 0..cnt-1. */
int catan_state_export(catan_env_t* env, int32_t* blob, const int64_t* env_idx, int64_t cnt, catan_stream_t stream);
int catan_state_import(catan_env_t* env, const int32_t* blob, const int64_t* env_idx, int64_t cnt, catan_stream_t stream);

/* env.reward_annealing_factor = f (RL/ppo/game_manager.py:164-166) */
int catan_set_reward_annealing(catan_env_t* env, double factor);
/* Rewards are computed in double exactly as env/wrapper.py:85-112 does (Python floats) and rounded ONCE to the float
 * `reward` buffer of catan_step.  A caller that accumulates rewards over several steps before rounding, as
 * RL/ppo/game_manager.py:94-95 does, registers a DEVICE double [n][4] here: every later step also stores the unrounded
 * rewards there (NULL: off, the default). */
int catan_set_reward_f64_buffer(catan_env_t* env, double* reward64);
/* number of rejected (mask-illegal) actions since creation; synchronises the stream.  The reference raises
 * RuntimeError on the first one (env/wrapper.py:38-41). */
int64_t catan_invalid_action_count(catan_env_t* env, catan_stream_t stream);

/* bench helper: `steps` x (sample_random_actions -> step) on one stream with the handle's own scratch buffers.
 * total_done (int64 device scalar, may be NULL) accumulates finished games. */
int catan_random_rollout(catan_env_t* env, uint32_t step_idx0, int64_t steps, catan_stream_t stream);

/* Deferred rollout: `iters` iterations of sample + step in which a game whose step needs the slow path (longest-road
 * recomputation after a road / settlement, re-deal after a win) stays BUSY - takes no action and draws nothing from its
 * policy stream - until the end of the current window of `window` iterations, when the slow path runs once for all of
 * them.  Each game draws its policy words with its OWN decision counter (catan_policy_counters) instead of a global step
 * index, so every game's trajectory is exactly the one the lock-step loop produces for that game after the same number of
 * decisions (tests/test_gpu_env_parity.py); only the interleaving between games differs.  Returns with every game idle
 * (the last window is closed).  Env-steps executed = sum of the counter increments. */
int catan_random_rollout_deferred(catan_env_t* env, int64_t iters, int32_t window, catan_stream_t stream);
/* per-game decision counters of the deferred rollout: read into / set from a DEVICE uint32[n] (in == NULL: zero them) */
int catan_policy_counters(catan_env_t* env, uint32_t* out, catan_stream_t stream);
int catan_set_policy_counters(catan_env_t* env, const uint32_t* in, catan_stream_t stream);

/* EnvWrapper._get_obs(): env/wrapper.py:52-83 (+ _get_tile_features :491-524, _get_player_inputs :526-709), batched.
 * out_f: float32 [n][1787] in the order proposed_trade[12] current_resources[6] tile_representations[19][60]
 * current_player_main[152] next/next_next/next_next_next_player_main[159]; out_lists: int32 [n][5][25] card-id lists
 * (current played, current hidden, next x3 played; ids = card+1, zero padded); out_lens: int32 [n][5] (1 when empty,
 * like the reference's [0]).  The deciding player of each game is catan_deciding_seat(). */
int catan_obs(catan_env_t* env, float* out_f, int32_t* out_lists, int32_t* out_lens, catan_stream_t stream);
/* The same observations in ONE pass for the rollout collector (RL/ppo/game_manager.py:69-140 keeps, per game, the list of the
 * ACTIVE seat's observations while every seat's policy needs the current one): `dense_*` = the [n][1787] matrix (float32, or
 * bfloat16 when bf16 != 0 - every value is a multiple of 1/8 below 32: exact) + int32 lists / lengths for the policy pass;
 * `rows_*` = the rollout storage, obs_f [steps][n][1787] of the same element type and int8 [steps][n][5][25] / [steps][n][5]:
 * game g with sel[g] != 0 appends its observation at step t_idx[g] (int64 [n], in range for the selected games).  Either
 * group may be NULL.  Replaces catan_obs + a cast + catan_masked_row_store of round 2. */
int catan_obs_rows(catan_env_t* env, int32_t bf16, void* dense_f, int32_t* dense_lists, int32_t* dense_lens, void* rows_f, int8_t* rows_lists,
                   int8_t* rows_lens, const int64_t* t_idx, const uint8_t* sel, catan_stream_t stream);

/* catan_obs_rows / catan_masks for a LIST of games (a collector that no longer evaluates its finished games: dense row j = the
 * observation / masks of game games[j], int32 [n_rows], n_rows <= n; a negative id = an unused row).  The rollout-storage rows
 * (rows_*, t_idx, sel: indexed by GAME as in catan_obs_rows) are written for the listed games only.  While a deferred sequence is
 * open, the mask row of a waiting game is a placeholder (only EndTurn legal) in catan_masks and catan_masks_of alike. */
int catan_obs_rows_of(catan_env_t* env, int32_t bf16, void* dense_f, int32_t* dense_lists, int32_t* dense_lens, void* rows_f, int8_t* rows_lists,
                      int8_t* rows_lens, const int64_t* t_idx, const uint8_t* sel, const int32_t* games, int64_t n_rows, catan_stream_t stream);
int catan_masks_of(catan_env_t* env, float* out_masks, const int32_t* games, int64_t n_rows, catan_stream_t stream);

/* Game.get_longest_path(player): game/game.py:843-862 for players[i] (PlayerId) in game i -> out[i].  Diagnostic/test
 * entry; inside catan_step the same search runs as part of update_longest_road. */
int catan_longest_path(catan_env_t* env, const int32_t* players, int32_t* out, catan_stream_t stream);

/* BatchProcessor.compute_advantages_alt: RL/ppo/process_batch.py:134-141.  rewards [T][N], values [T+1][N]
 * (denormalised), masks [T+1][N] -> returns [T][N], adv_raw [T][N] = returns - values[:-1], and
 * stats3 (device double[3]) = (sum, sum of squares, count) of adv_raw for the global normalisation.
 * workspace: device double[catan_gae_workspace_doubles(N)]. */
int64_t catan_gae_workspace_doubles(int64_t N);
int catan_gae(const float* rewards, const float* values, const float* masks, int64_t T, int64_t N, double gamma, double lam,
              float* returns, float* adv_raw, double* workspace, double* stats3, catan_stream_t stream);
/* process_batch.py:142: adv = (adv - mean) / (std_unbiased + 1e-5) with (sum, sumsq, count) in stats3 - all-reduce
 * the three doubles over ranks first when games are sharded over GPUs. */
int catan_adv_normalise(float* adv, int64_t total, const double* stats3, catan_stream_t stream);

/* PPO.update loss: RL/ppo/ppo.py:46-48 (value normaliser, when use_norm) and :54-66.  All arrays float32 [B] on device.
 * losses2 = (action_loss, value_loss); d_logp, d_values = gradient of value_coef*value_loss + action_loss w.r.t.
 * action_log_probs and values (the entropy term stays in the network's autograd graph).  workspace: device double
 * [catan_ppo_loss_workspace_doubles()], ZERO before the first call and left zero by every call (per-workgroup partial sums
 * and an arrival counter: the kernel runs on up to 256 workgroups and adds the partial sums up in a fixed order). */
int64_t catan_ppo_loss_workspace_doubles(void);
int catan_ppo_loss(const float* logp, const float* old_logp, const float* adv, const float* values, const float* old_values,
                   const float* returns, int64_t B, float clip, float value_coef, int use_norm, float norm_mean, float norm_std,
                   float* losses2, float* d_logp, float* d_values, double* workspace, catan_stream_t stream);

/* Game.randomise_uncertainty(controlling_player_id): game/game.py:1207-1282 (forward search, worker.py:46): re-deals
 * the dev-card pile + the other players' hidden cards and the other players' resource hands, consistently with what the
 * controlling player knows (his opponent_min_res / opponent_max_res bounds, hand sizes, the bank).  controlling_player:
 * DEVICE int32[n], PlayerId 1..4 per game, 0 = leave that game alone.  Draws from the game's own RNG stream; the packed
 * masks are recomputed.  The reference's rejection loop does not terminate on a state whose bounds admit no consistent
 * deal (it only ever calls this on freshly restored states); here the loop is capped at 100 000 attempts and such games
 * are counted by catan_inconsistent_deal_count. */
int catan_randomise_uncertainty(catan_env_t* env, const int32_t* controlling_player, catan_stream_t stream);
int64_t catan_inconsistent_deal_count(catan_env_t* env, catan_stream_t stream);

/* Not part of the drop-in boundary, declared in their own headers (same library):
 *   catan_hip_nn.h      the policy net's hand-written kernels (RL/models: attention, LayerNorm, tall-skinny linears, the
 *                       tile encoder, the action heads, the dev-card modules, the LSTM cell, the masked categorical)
 *   catan_hip_tuning.h  scheduling knobs, counters and profilers of the env kernels (benchmarks and diagnostics only) */


#ifdef __cplusplus
}
#endif
#endif
