// catan_collector.hip - the per-game bookkeeping of rollout collection (reference RL/ppo/game_manager.py:69-140) for all games of
// a batched env, one lane per game, two launches per env iteration instead of ~40 small tensor operations.
//
// The reference keeps, per game, Python lists of the ACTIVE seat's observations / actions / action masks / log-probs / rewards /
// terminal masks and appends to them as the four seats take their turns; rollout.RolloutCollector keeps four counters per game
// (n_obs, n_msk, n_act, n_rew) and writes the rollout tensors directly in the (T + 1, N, ...) layout of process_batch.py:37-104.
// k_collector_pre: the actions the env is stepped with (a frozen game - one that already holds its T + 1 observations - gets the
// no-op) and the `live` flags.  k_collector_post: everything after env.step (:91-136), including which games append their next
// observation and where (sel / t_obs for catan_obs_rows).  Line references are to game_manager.py.
#pragma once

namespace catan {

struct CollectorArgs {
    long n; int T;
    // per-game state of the collector
    long long* n_obs; long long* n_msk; long long* n_act; long long* n_rew;   // int64 [n]
    double* racc;                    // [n][4]: rewards summed over the seats' moves since the active seat's last decision (:94-95)
    u8* done_since;                  // [n]: bool
    float* term;                     // [n]: the terminal mask the policy sees next (:97)
    u8* pending_obs;                 // [n]: bool: the observation of the next iteration belongs to the active seat (:126-133)
    u8* live;                        // [n]: written by the pre kernel
    u8* sel; long long* t_obs;       // [n]: which games append their next observation, and at which step (for catan_obs_rows)
    const long long* active_pid;     // int64 [n]: PlayerId 1..4 of the active seat
    // this iteration
    const i32* deciding;             // int32 [n]: deciding player BEFORE the step
    const i32* n_deciding;           // int32 [n]: deciding player AFTER the step (and the auto-reset)
    const long long* actions;        // int64 [n][18]
    const float* logp;               // [n]
    const i32* pmasks;               // int32 [n][11]: packed masks of the state the action was chosen in
    const float* reward; const double* reward64;   // [n][4]; reward64 may be null
    const u8* done;                  // [n]
    // rollout storage
    long long* st_actions; float* st_logp; i32* st_amasks;   // [T][n][18], [T][n], [T][n][11]
    float* st_rewards; float* st_masks;                      // [T + 2][n]
    long long* n_complete;           // [1]: finished games, accumulated
    i32* a_env;                      // pre: int32 [n][18] for catan_step
    // catan_step_deferred (both null: catan_step - every game's step is complete when the post kernel runs)
    const u8* waiting_before;        // [n]: the game was waiting when the step was called (its action was ignored)
    const u8* status;                // [n]: CATAN_STEP_COMPLETE 0 / CATAN_STEP_WAITING 1 after the call
};

__global__ __launch_bounds__(256) void k_collector_pre(CollectorArgs a) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n) return;
    const bool live = a.n_obs[g] < a.T + 1;                                  // while len(observations) < T + 1 (:78)
    a.live[g] = live ? 1 : 0;
    const long long* src = a.actions + g * 18;
    i32* dst = a.a_env + g * 18;
#pragma unroll
    for (int k = 0; k < 18; k++) dst[k] = (i32)src[k];
    if (!live) dst[0] = -1;                                                  // frozen games: the env's explicit no-op
}

__global__ __launch_bounds__(256) void k_collector_post(CollectorArgs a) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const bool in = g < a.n;
    bool done = false;
    if (in) {
        const int T = a.T;
        const long n = a.n;
        const bool live = a.live[g] != 0;
        // Under the deferred schedule the two halves of the bookkeeping of one decision may run in different iterations: the append
        // of the action (:102-105) when the env CONSUMES it, everything that needs the step's result (:91-97, :106-136) when the env
        // DELIVERS it (the game waited in between: it took no other action and appended nothing).  catan_step: both at once.
        const bool consumed = live && (a.waiting_before == nullptr || a.waiting_before[g] == 0);
        const bool delivered = live && (a.status == nullptr || a.status[g] == 0);
        const int ap = (int)a.active_pid[g];
        long long n_act = a.n_act[g];
        if (consumed && a.deciding[g] == ap) {                               // :102-105
            const long t = n_act < T - 1 ? n_act : T - 1;
            const long long* src = a.actions + g * 18;
            long long* dst = a.st_actions + (t * n + g) * 18;
#pragma unroll
            for (int k = 0; k < 18; k++) dst[k] = src[k];
            a.st_logp[t * n + g] = a.logp[g];
            const i32* ms = a.pmasks + g * 11;
            i32* md = a.st_amasks + (t * n + g) * 11;
#pragma unroll
            for (int k = 0; k < 11; k++) md[k] = ms[k];
            n_act += 1;
            a.n_act[g] = n_act;
        }
        bool sel = false;
        if (delivered) {
            done = a.done[g] != 0;
            a.term[g] = done ? 0.0f : 1.0f;                                  // :97
            double r[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                r[k] = a.racc[g * 4 + k] + (a.reward64 != nullptr ? a.reward64[g * 4 + k] : (double)a.reward[g * 4 + k]);   // :94-95
            const bool next_active = a.n_deciding[g] == ap;
            bool done_since = a.done_since[g] != 0;
            // :106-110 (not done: uses the post-step deciding player) and :112-118 (done: exactly one reward is appended)
            const bool app = done ? true : (next_active && n_act > 0 && !done_since);
            if (app) {
                long long n_rew = a.n_rew[g];
                const long t = n_rew < T + 1 ? n_rew : T + 1;
                a.st_rewards[t * n + g] = (float)r[ap - 1];                  // process_batch.py:63: one rounding
                a.n_rew[g] = n_rew + 1;
                r[ap - 1] = 0.0;
            }
            long long n_msk = a.n_msk[g];
            if (done) {                                                      // :112-124
                const long t = n_msk < T + 1 ? n_msk : T + 1;
                a.st_masks[t * n + g] = 0.0f;
                n_msk += 1;
                done_since = false;
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = 0.0;
            }
            const bool add_mask = next_active && !done && !done_since;       // :128-136
            if (add_mask) {
                const long t = n_msk < T + 1 ? n_msk : T + 1;
                a.st_masks[t * n + g] = 1.0f;
                n_msk += 1;
            }
            a.n_msk[g] = n_msk;
            done_since = next_active ? false : (done ? true : done_since);
            a.done_since[g] = done_since ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 4; k++) a.racc[g * 4 + k] = r[k];
            a.pending_obs[g] = next_active ? 1 : 0;
            // the next iteration's observation append (catan_obs_rows): the active seat's, while the game still misses observations
            const long long n_obs = a.n_obs[g];
            sel = next_active && n_obs < T + 1;
            a.t_obs[g] = n_obs < T ? n_obs : T;
            if (sel) a.n_obs[g] = n_obs + 1;
        } else {
            a.pending_obs[g] = 0;
            a.t_obs[g] = 0;
        }
        a.sel[g] = sel ? 1 : 0;
    }
    const unsigned long long b = __ballot(done);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(reinterpret_cast<unsigned long long*>(a.n_complete), (unsigned long long)__popcll(b));
}

}  // namespace catan
