"""Batched on-device rollout collection with the reference's active-seat bookkeeping.

Replaces `GamesAndPoliciesManager.gather_rollouts` / `_after_rollouts` / `reset` (reference RL/ppo/game_manager.py:35-59,
69-150) and the worker/pipe layer above it (RL/ppo/vec_gather_experience.py) by one lock-step device loop over all games:
every iteration encodes the observation of each game's deciding player (k_obs), runs the policy of that seat, steps
all games (k_step) and updates the per-game bookkeeping with vectorised tensor ops.  Only the decisions of each game's
ACTIVE seat (the seat mapped to the central policy) are stored; rewards are accumulated across the other seats' moves.

The bookkeeping restates game_manager.py line by line (including its quirks, see comments) with four per-game counters
instead of Python lists: n_obs (observations), n_msk (terminal masks), n_act (actions / log-probs / action masks) and
n_rew (rewards).  A game is frozen (no-op actions) once it holds T+1 observations, until the slowest game catches up.
Storage is written directly in the `(T+1, N, ...)` layout of `BatchProcessor.process_rollouts`
(RL/ppo/process_batch.py:37-104), so no restacking is needed.

The env is duck-typed (`n`, `device`, `deciding_player`, `get_obs`, `get_action_masks`, `step` with auto-reset and
no-op for type < 0): `env.VecCatanEnv` on the GPU; the CPU tests use an oracle-backed stand-in.
"""
import torch

from . import spec


class RolloutStorage(object):
    """The tensors `BatchProcessor` holds after `process_rollouts` (process_batch.py:37-104), device resident."""
    _tokens = iter(range(1, 1 << 62))

    def invalidate(self):
        """the contents changed by other means than gather_rollouts (a loaded rollout, an in-place edit): consumers that cache
        per-rollout derived data (PPOTrainer's distinct boards) key on (token, generation) and recompute"""
        self.generation += 1

    def __init__(self, T, N, device, obs_dtype=torch.float32, lstm_size=0):
        self.T, self.N = T, N
        self.token = next(RolloutStorage._tokens)      # process-unique: a later storage at the same address is another rollout
        # process_batch.py:53-59: the active seat's LSTM state (h, c) entering each of its stored decisions
        self.hidden = torch.zeros((2, T + 1, N, lstm_size), dtype=torch.float32, device=device) if lstm_size else None
        self.obs_f = torch.zeros((T + 1, N, spec.OBS_FLOATS), dtype=obs_dtype, device=device)
        self.lists = torch.zeros((T + 1, N, 5, spec.OBS_LIST_PAD), dtype=torch.int8, device=device)
        self.lens = torch.ones((T + 1, N, 5), dtype=torch.int8, device=device)
        self.masks = torch.ones((T + 2, N), dtype=torch.float32, device=device)      # terminal masks (one spare slot, see reset quirk)
        self.rewards = torch.zeros((T + 2, N), dtype=torch.float32, device=device)
        self.actions = torch.zeros((T, N, spec.ACTION_WORDS), dtype=torch.int64, device=device)
        self.action_log_probs = torch.zeros((T, N), dtype=torch.float32, device=device)
        self.action_masks = torch.zeros((T, N, 11), dtype=torch.int32, device=device)   # packed 325-bit masks
        self.games_complete = 0
        self.generation = 0           # bumped by every gather_rollouts (consumers cache per-rollout derived data on it)

    def unpack_action_masks(self, packed):
        """int32 [..., 11] -> float32 [..., 325]"""
        if packed.is_cuda and packed.dtype == torch.int32:            # one kernel (catan_expand_masks) instead of shift / and / slice / cast passes
            import ctypes as C
            from . import _lib
            p = packed.contiguous()
            rows = p.numel() // p.shape[-1]
            out = torch.empty(packed.shape[:-1] + (spec.MASK_WORDS,), dtype=torch.float32, device=p.device)
            if rows == 0:                                             # an empty selection (empty minibatch / group): nothing to expand
                return out
            _lib.check(_lib.lib().catan_expand_masks(C.c_void_p(p.data_ptr()), rows, int(p.shape[-1]), C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return out
        bits = (packed[..., None] >> torch.arange(32, device=packed.device, dtype=torch.int32)) & 1
        return bits.reshape(packed.shape[:-1] + (352,))[..., :spec.MASK_WORDS].float()


def pack_action_masks(m):
    """float [N,325] -> int32 [N,11] (bit i of the flat mask -> word i>>5, bit i&31)"""
    N = m.shape[0]
    b = torch.zeros((N, 352), dtype=torch.int64, device=m.device)
    b[:, :spec.MASK_WORDS] = (m > 0).long()
    w = (b.reshape(N, 11, 32) << torch.arange(32, device=m.device)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).int()


class RolloutCollector(object):
    GRAPH_ACT_MIN_GAMES = 8192

    def __init__(self, env, policy, num_steps, opponents=None, seed=0, autocast_dtype=None, graph_act=None, deferred_window=None, act_buckets=None):
        """policy: central net (policy 0); opponents: list of up to 3 nets for policy slots 1..3 of every game (None =
        every seat plays the central policy).  A league (league.League.assign) installs per-game opponents instead.
        deferred_window: step the env with catan_step_deferred (window of that many iterations) instead of catan_step - the
        reference's workers advance every env independently (game_manager.py:78-113), and so do the games here: one whose step
        needs the slow path waits for it while the others go on (device collector only; 0 = catan_step; None = 4 where the env
        has the call: measured at 65 536 games x T = 200, tools/rollout_schedules.py: 2.50 s with catan_step, 2.44 s with W = 4).
        act_buckets: row counts of the captured policy passes (graph_act): once the games that still miss observations fit a
        smaller bucket, only they are evaluated (None = N, N/2, ... N/16 with graph_act, else N only)."""
        self.env, self.policy, self.T = env, policy, num_steps
        self.deferred_window = self.DEFAULT_DEFERRED_WINDOW if deferred_window is None else int(deferred_window)
        self.act_buckets = None if act_buckets is None else tuple(sorted(set(int(b) for b in act_buckets) | {env.n}))
        self.N, self.device = env.n, env.device
        if torch.device(self.device).type == "cuda":
            from . import nn_kernels
            nn_kernels.use_tuned_gemms()
        self.autocast_dtype = autocast_dtype
        self.opponent_nets, self.opp_index, self._graphed_nets = [], None, {}
        if opponents:
            nets = list(opponents)
            self.set_opponents(nets, torch.tensor([[min(j, len(nets) - 1) for j in range(3)]]).expand(self.N, 3))
        # acting nets under autocast hold their weights in the autocast dtype (policy.inference_copy): no per-call casts
        self._shadow = policy.inference_copy(autocast_dtype) if (autocast_dtype is not None and hasattr(policy, "inference_copy")) else None
        g = torch.Generator(device="cpu").manual_seed(seed)
        # game_manager.py:24-31: a random seat order per game; order[0] is the active player, order[j] plays policy j
        perm = torch.stack([torch.randperm(4, generator=g) for _ in range(self.N)])        # [N,4] pid0 per policy slot
        self.policy_of_pid = torch.empty((self.N, 4), dtype=torch.int64)
        self.policy_of_pid.scatter_(1, perm, torch.arange(4).expand(self.N, 4))
        self.policy_of_pid = self.policy_of_pid.to(self.device)
        self.active_pid = (perm[:, 0] + 1).long().contiguous().to(self.device)                                  # PlayerId 1..4
        self.sample_gen = torch.Generator(device=self.device).manual_seed(seed + 1)
        self.recurrent = bool(getattr(policy, "include_lstm", False))
        # Self-play with one feed-forward net: the policy pass of an env iteration is ~500 small launches and host-bound
        # (7.0 ms of host time for 5.3 ms of kernels at 65 536 rows) - replayed as one captured hipGraph instead
        # (forward_search.GraphedAct; sampling draws from the same registered generator).  graph_act: None = automatic.
        if graph_act is None:
            graph_act = (torch.device(self.device).type == "cuda" and not self.recurrent and hasattr(policy, "refresh_kernel_packs")
                         and self.N >= self.GRAPH_ACT_MIN_GAMES)
        self.graph_act, self._graphed = bool(graph_act), None
        self.lstm_size = int(policy.lstm_size) if self.recurrent else 0
        # Under autocast the net casts its inputs to the autocast dtype before the first GEMM anyway, and every observation
        # value is a small multiple of 1/8 (exact in bf16): the rollout tensors are kept in that dtype - half the HBM
        # (config 3: 58 instead of 116 GB) and no cast pass per minibatch.
        obs_dtype = autocast_dtype if (autocast_dtype in (torch.bfloat16, torch.float16) and torch.device(self.device).type == "cuda") else torch.float32
        self.storage = RolloutStorage(num_steps, self.N, self.device, obs_dtype=obs_dtype, lstm_size=self.lstm_size)
        # game_manager.py:94-95 sums the env's rewards (Python floats) over the other seats' moves and process_batch.py:63
        # rounds the sum to fp32: the env leaves its unrounded rewards in a float64 buffer for that
        self.reward64 = env.enable_reward64() if hasattr(env, "enable_reward64") else None
        self.reset()

    def set_opponents(self, nets, opp_index):
        """nets: the distinct opponent nets in play; opp_index int64 [N,3]: which of them plays policy slots 1..3 of
        each game (game_manager.py:15,28-31: the slot -> seat map of a game stays fixed)."""
        self._graphed_nets = {}                  # (captured per-net passes belong to the nets they were captured with)
        self.opponent_nets = [n.inference_copy(self.autocast_dtype) if (getattr(self, "autocast_dtype", None) is not None and hasattr(n, "inference_copy")
                                                                        and getattr(n, "_inference_dtype", None) is None) else n for n in nets]
        self.opp_index = opp_index.to(self.device).long().contiguous() if len(self.opponent_nets) else None

    # game_manager.py:35-59 (the env itself is already reset: EnvWrapper.reset() happened in catan_create / env.reset())
    def reset(self):
        N, dev, st = self.N, self.device, self.storage
        # the four counters are rows of one tensor (and only ever updated in place): catan_collector_post takes them as one block
        self._cnt = torch.zeros((4, N), dtype=torch.int64, device=dev)
        self.n_obs, self.n_msk, self.n_act, self.n_rew = self._cnt[0], self._cnt[1], self._cnt[2], self._cnt[3]
        self.n_msk.fill_(1)                                                                 # terminal_masks = [1.0]
        self._flags = torch.zeros((4, N), dtype=torch.uint8, device=dev)                    # done_since, pending_obs, live, sel of the fused bookkeeping
        st.masks[0] = 1.0
        self.pending_obs = self.env.deciding_player().long() == self.active_pid             # observations = [obs] iff the active seat moves first
        self.done_since = torch.zeros(N, dtype=torch.bool, device=dev)
        self.racc = torch.zeros((N, 4), dtype=torch.float64, device=dev)
        if self.recurrent:            # game_manager.py:54-59: every seat of every game starts from the zero state
            self.hid = torch.zeros((2, N, 4, self.lstm_size), dtype=torch.float32, device=dev)

    # ---- "append to this game's list" for all games at once.  The selected games are never compacted on the host (a
    # `nonzero` is a device-to-host read of how many there are: five of them per env iteration kept the host from running
    # ahead, and the GPU idle while the host launched the next policy pass): big rows go through catan_masked_row_store,
    # per-game scalars through a read-modify-write of the whole column.
    def _row_store(self, dst, src, t, sel):
        """dst[t[n], n] = src[n] where sel[n]; dst [steps, N, ...], src [N, ...] (same trailing shape and dtype)"""
        if dst.is_cuda and hasattr(self.env, "L"):
            import ctypes as C
            from . import _lib
            src = src.contiguous()
            row_bytes = src[0].numel() * src.element_size()
            _lib.check(_lib.lib().catan_masked_row_store(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(t.data_ptr()),
                                                         C.c_void_p(sel.data_ptr()), self.N, row_bytes, dst.stride(0) * dst.element_size(),
                                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return
        idx = sel.nonzero(as_tuple=True)[0]
        if idx.numel():
            dst[t[idx], idx] = src[idx]

    def _col_store(self, dst, value, t, sel):
        """dst[t[n], n] = value[n] (or a scalar) where sel[n]; t in range for every n"""
        ar = self._ar
        cur = dst[t, ar]
        v = value if torch.is_tensor(value) else torch.full_like(cur, value)
        m = sel if cur.dim() == 1 else sel.reshape((-1,) + (1,) * (cur.dim() - 1))
        dst[t, ar] = torch.where(m, v.to(cur.dtype), cur)

    def _store_obs(self, sel, f, lists, lens, t=None):
        """f is None: the observation rows were already appended by the env (catan_obs_rows); only the bookkeeping is left"""
        st, T = self.storage, self.T
        if t is None:
            t = self.n_obs.clamp(max=T)
        sel8 = sel.to(torch.uint8)
        if f is not None:
            self._row_store(st.obs_f, f.to(st.obs_f.dtype), t, sel8)
            self._row_store(st.lists, lists.to(torch.int8), t, sel8)
            self._col_store(st.lens, lens.to(torch.int8), t, sel)
        if self.recurrent:            # game_manager.py:55,133: the state the active seat will enter this decision with
            hid = self.hid[:, self._ar, self.active_pid - 1]                                  # [2, N, L]
            for k in range(2):
                self._row_store(st.hidden[k], hid[k], t, sel8)
        self.n_obs += sel.long()

    fused_bookkeeping = True   # False: the tensor-operation form of the bookkeeping below (what the kernels are tested against)
    CHECK_EVERY = 8      # (tensor-operation form) env iterations between two host reads of "every game has its T + 1 observations"
    DEFAULT_DEFERRED_WINDOW = 4
    LIVE_LAG = 2         # (device collector) the host looks at the live-game count of this many iterations ago: no host wait per iteration

    def _group_buckets(self):
        """row counts of the captured per-net passes of a league rollout: a net's share of the games is about a quarter"""
        N = self.N
        return tuple(sorted({max(1024, N * k // 16) for k in (1, 2, 3, 4, 5, 6, 8, 12, 16)}))

    def _bucket_list(self):
        if self.act_buckets is not None:
            return self.act_buckets
        N = self.N
        if not self.graph_act:
            return (N,)
        halves = {N >> k for k in range(0, 6) if (N >> k) >= 1024}
        return tuple(sorted(halves | {3 * (b >> 2) for b in halves if 3 * (b >> 2) >= 1024}))      # N, 3N/4, N/2, 3N/8, ...

    @torch.no_grad()
    def gather_rollouts(self, max_iters=None):
        """game_manager.py:69-140.  Returns the storage (first T(+1) entries per game are the rollout)."""
        try:
            return self._gather_rollouts(max_iters)
        except BaseException:
            # an error inside the loop (a policy that raises, out of memory) must not leave a catan_step_deferred sequence open:
            # every later step / reset / export of the env would be refused until someone flushed it
            flush = getattr(self.env, "step_flush", None)
            if flush is not None and self.deferred_window:
                try:
                    flush()
                except Exception:
                    pass
            raise

    def _gather_rollouts(self, max_iters=None):
        env, st, T, N, dev = self.env, self.storage, self.T, self.N, self.device
        ar = self._ar = torch.arange(N, device=dev)
        if self._shadow is not None:
            self._shadow.load_from(self.policy)  # the central policy as of this rollout (game_manager.py:161-162 `_update_policy`)
        self.racc.zero_()                        # `rewards = {...: 0}` at the start of every gather call (:76)
        self.done_since.zero_()                  # `done_since_prev_turn = [False ...]` (:77)
        term = st.masks[0].clone()               # `terminal_mask = terminal_masks[env_num][0]` (:74-75)
        iters = 0
        n_live_iters = torch.zeros((), dtype=torch.int64, device=dev)       # iterations in which some game still stepped
        n_complete = torch.zeros((), dtype=torch.int64, device=dev)
        packed_from_env = hasattr(env, "get_action_masks_packed")
        # one kernel writes the dense observations the policy pass reads (in the storage's dtype: every value is exact in bf16)
        # AND appends the active seats' rows to the storage (k_obs_rows; round 2: k_obs, a cast pass and a masked row store)
        fused_obs = hasattr(env, "get_obs_rows") and st.obs_f.is_cuda and st.obs_f.dtype in (torch.float32, torch.bfloat16)
        obs_out = mask_out = None
        # ... and two kernels do the per-game bookkeeping of an iteration (catan_collector_pre / _post) instead of ~40 tensor operations
        fused_book = fused_obs and packed_from_env and not self.recurrent and hasattr(env, "L") and self.fused_bookkeeping
        if fused_book:
            import ctypes as C
            from . import _lib
            L, P = _lib.lib(), (lambda x: C.c_void_p(x.data_ptr()))
            fl = self._flags
            fl[0].copy_(self.done_since)
            fl[1].copy_(self.pending_obs)
            t_next = torch.empty(N, dtype=torch.int64, device=dev)
            a_env = torch.empty((N, spec.ACTION_WORDS), dtype=torch.int32, device=dev)
            live8, sel_next = fl[2], fl[3]
            first = True
            deferred = bool(self.deferred_window) and hasattr(env, "step_deferred")
            if deferred:             # status of the previous / of this catan_step_deferred call (alternating)
                stat, sk = torch.zeros((2, N), dtype=torch.uint8, device=dev), 0
            # Only the games that still miss observations are evaluated once they fit a smaller captured policy pass: `games` is
            # the list of those games as of the last bucket change (a superset of them afterwards: a game that has frozen since
            # just gets the no-op), row j of the policy pass is game games[j].
            buckets = self._bucket_list() if ((self.graph_act or self.act_buckets is not None) and not self.opponent_nets and not self.recurrent) else (N,)
            B, games, games_l, cnt = N, None, None, N
            act_full = logp_full = None
            RING = self.LIVE_LAG + 2
            live_pin = torch.empty(RING, dtype=torch.int64).pin_memory() if dev != "cpu" and torch.device(dev).type == "cuda" else torch.empty(RING, dtype=torch.int64)
            live_ev = [torch.cuda.Event() for _ in range(RING)]
            live_bound = N
            self.bucket_log = []         # (iteration, rows of the policy pass, listed games) at every bucket change of this rollout
        while True:
            if fused_book:
                if first:
                    sel = self.pending_obs & (self.n_obs < T + 1)
                    t_obs = self.n_obs.clamp(max=T)
                else:
                    sel, t_obs = sel_next, t_next       # written by catan_collector_post, which also counted the appends in n_obs
                # the live-game count of LIVE_LAG iterations ago (an upper bound of today's: frozen games stay frozen)
                j = iters - self.LIVE_LAG
                if j >= 1:
                    live_ev[j % RING].synchronize()
                    live_bound = int(live_pin[j % RING])
                done_all = live_bound == 0
                newB = next(b for b in buckets if b >= max(live_bound, 1))
                if newB < B and not done_all:
                    live_now = self.n_obs < T + 1       # (host read: at most len(buckets) - 1 times per rollout)
                    games_l = live_now.nonzero(as_tuple=True)[0]
                    # a game that froze since the count was taken would lose the append of its last observation if it were dropped
                    # from the list before this iteration's catan_obs_rows: keep the selected ones
                    games_l = (live_now | sel.bool()).nonzero(as_tuple=True)[0]
                    if games_l.numel() <= newB:
                        games, cnt, B = games_l.to(torch.int32).contiguous(), int(games_l.numel()), newB
                        self.bucket_log.append((iters, B, cnt))
                        obs_out = mask_out = None
                        if act_full is None:
                            act_full = torch.zeros((N, spec.ACTION_WORDS), dtype=torch.int64, device=dev)
                            logp_full = torch.zeros((N,), dtype=torch.float32, device=dev)
                oo = None if obs_out is None else tuple(x[:cnt] for x in obs_out)
                f, lists, lens = env.get_obs_rows(st.obs_f.dtype, out=oo, rows=(st.obs_f, st.lists, st.lens), t=t_obs, sel=sel, games=games) \
                    if games is not None else env.get_obs_rows(st.obs_f.dtype, out=obs_out, rows=(st.obs_f, st.lists, st.lens), t=t_obs, sel=sel)
                if first:
                    self.n_obs += sel.long()
                    first = False
                if max_iters is not None and iters >= max_iters:
                    break
                if done_all:
                    break
                iters += 1
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                deciding = env.deciding_player()                                                # :79
                if games is not None:
                    masks = env.get_action_masks(None if mask_out is None else mask_out[:cnt], games=games)
                else:
                    masks = env.get_action_masks(mask_out) if mask_out is not None else env.get_action_masks()   # :83
                pol = self.policy_of_pid[ar, deciding.long() - 1] if self.opponent_nets else None
                actions, logp = self._act(f, lists, lens, masks, pol, games=games)              # :85-89
                if obs_out is None and self._graphed is not None and not self.opponent_nets:
                    bufs = self._graphed.static_inputs(B)
                    if bufs is not None and bufs[0].dtype == st.obs_f.dtype and bufs[1].dtype == torch.int32 and bufs[2].dtype == torch.int32:
                        obs_out, mask_out = bufs[:3], bufs[3]
                if games is not None:                   # back to one row per game (the games outside the list are frozen: no-ops)
                    act_full.index_copy_(0, games_l, actions)
                    logp_full.index_copy_(0, games_l, logp)
                    actions, logp = act_full, logp_full
                actions, logp = actions.contiguous(), logp.contiguous()
                _lib.check(L.catan_collector_pre(N, T, P(self.n_obs), P(actions), P(a_env), P(live8), stream))
                n_live_iters += live8.any()
                pmasks = env.get_action_masks_packed()                                          # (before the step replaces them)
                if deferred:
                    wb, so = stat[sk], stat[sk ^ 1]
                    reward, done, _ = env.step_deferred(a_env, self.deferred_window, status_out=so)
                    sk ^= 1
                else:
                    wb = so = None
                    reward, done = env.step(a_env)                                              # :91 (auto-reset == :113)
                n_deciding = env.deciding_player()
                _lib.check(L.catan_collector_post(N, T, P(self._cnt), P(self.racc), P(fl), P(term), P(t_next), P(self.active_pid), P(deciding), P(n_deciding),
                                                  P(actions), P(logp), P(pmasks), P(reward), P(self.reward64) if self.reward64 is not None else None, P(done),
                                                  P(st.actions), P(st.action_log_probs), P(st.action_masks), P(st.rewards), P(st.masks), P(n_complete),
                                                  P(wb) if deferred else None, P(so) if deferred else None, stream))
                slot = iters % RING
                live_pin[slot].copy_((self.n_obs < T + 1).sum(), non_blocking=True)
                live_ev[slot].record()
                continue
            if fused_obs:
                sel = self.pending_obs & (self.n_obs < T + 1)
                t_obs = self.n_obs.clamp(max=T)
                f, lists, lens = env.get_obs_rows(st.obs_f.dtype, out=obs_out, rows=(st.obs_f, st.lists, st.lens), t=t_obs, sel=sel)
                self._store_obs(sel, None, None, None, t=t_obs)
            else:
                f, lists, lens = env.get_obs()
                # an observation produced by the previous step for the active seat (:126-133) - or the carried one
                self._store_obs(self.pending_obs & (self.n_obs < T + 1), f, lists, lens)
            self.pending_obs = torch.zeros(N, dtype=torch.bool, device=dev)
            frozen = self.n_obs >= T + 1                                                    # while len(observations) < T+1 (:78)
            if max_iters is not None and iters >= max_iters:
                break
            if iters % self.CHECK_EVERY == 0 and bool(frozen.all()):
                break
            iters += 1
            live = ~frozen
            n_live_iters += live.any()
            deciding = env.deciding_player().long()                                         # :79
            masks = env.get_action_masks(mask_out) if mask_out is not None else env.get_action_masks()   # :83
            pol = self.policy_of_pid[ar, deciding - 1]
            actions, logp = self._act(f, lists, lens, masks, pol, deciding, term, live)     # :85-89
            if fused_obs and obs_out is None and self._graphed is not None and not self.opponent_nets:
                # from now on the env writes straight into the captured graph's input buffers (no copy per replay)
                bufs = self._graphed.static_inputs(N)
                if bufs is not None and bufs[0].dtype == st.obs_f.dtype and bufs[1].dtype == torch.int32 and bufs[2].dtype == torch.int32:
                    obs_out, mask_out = bufs[:3], bufs[3]
            a_env = actions.to(torch.int32)
            a_env[:, 0] = torch.where(frozen, torch.full_like(a_env[:, 0], -1), a_env[:, 0])   # frozen games: no-op
            pmasks = env.get_action_masks_packed() if packed_from_env else pack_action_masks(masks)   # (before the step replaces them)
            reward, done = env.step(a_env)                                                  # :91 (auto-reset == :113)
            done = done.bool() & live
            term = torch.where(live, 1.0 - done.float(), term)                              # :97
            self.racc += (reward.double() if self.reward64 is None else self.reward64) * live[:, None]   # :94-95
            was_active = (deciding == self.active_pid) & live                               # :102-105
            t = self.n_act.clamp(max=T - 1)
            self._col_store(st.actions, actions, t, was_active)
            self._col_store(st.action_log_probs, logp, t, was_active)
            self._col_store(st.action_masks, pmasks, t, was_active)
            self.n_act += was_active.long()
            n_deciding = env.deciding_player().long()                                       # after the step (and the reset)
            next_active = (n_deciding == self.active_pid) & live
            r_active = self.racc[ar, self.active_pid - 1]
            # :106-110 (not done: uses the post-step deciding player) and :112-118 (done: exactly one reward is appended)
            app = torch.where(done, torch.ones_like(done), next_active & (self.n_act > 0) & ~self.done_since) & live
            self._col_store(st.rewards, r_active.float(), self.n_rew.clamp(max=T + 1), app)        # process_batch.py:63
            self.n_rew += app.long()
            self.racc[ar, self.active_pid - 1] = torch.where(app, torch.zeros_like(r_active), r_active)
            # :112-124
            self._col_store(st.masks, 0.0, self.n_msk.clamp(max=T + 1), done)
            self.n_msk += done.long()
            self.done_since = self.done_since & ~done
            self.racc = torch.where(done[:, None], torch.zeros_like(self.racc), self.racc)
            if self.recurrent:
                self.hid = torch.where(done[None, :, None, None], torch.zeros_like(self.hid), self.hid)   # :121-124
            n_complete += done.sum()
            # :128-136
            add_mask = next_active & ~done & ~self.done_since
            self._col_store(st.masks, 1.0, self.n_msk.clamp(max=T + 1), add_mask)
            self.n_msk += add_mask.long()
            self.done_since = torch.where(next_active, torch.zeros_like(self.done_since),
                                          torch.where(done & live, torch.ones_like(self.done_since), self.done_since))
            self.pending_obs = next_active
        if fused_book:
            if deferred:
                # the steps that are still outstanding (none when every game has frozen; some after `max_iters`): completed by the
                # flush, their results booked as in the loop, the observations they make the active seat's appended
                reward, done, so = env.step_flush()
                ones = torch.ones(N, dtype=torch.uint8, device=dev)
                deciding = n_deciding = env.deciding_player()
                _lib.check(L.catan_collector_post(N, T, P(self._cnt), P(self.racc), P(fl), P(term), P(t_next), P(self.active_pid), P(deciding), P(n_deciding),
                                                  P(actions) if iters else P(a_env), P(logp) if iters else P(term), P(env.get_action_masks_packed()), P(reward),
                                                  P(self.reward64) if self.reward64 is not None else None, P(done),
                                                  P(st.actions), P(st.action_log_probs), P(st.action_masks), P(st.rewards), P(st.masks), P(n_complete),
                                                  P(ones), P(so), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                env.get_obs_rows(st.obs_f.dtype, rows=(st.obs_f, st.lists, st.lens), t=t_next, sel=sel_next, dense=False)
            # (the loop left after an observation append: sel / pending_obs of the flags are consumed)
            self.done_since = fl[0].bool()
            self.pending_obs = torch.zeros(N, dtype=torch.bool, device=dev)
        st.games_complete += int(n_complete)
        st.generation += 1
        self.iters = int(n_live_iters) if max_iters is None else iters
        return st

    def _act(self, f, lists, lens, masks, pol, deciding=None, term=None, live=None, games=None):
        """One batched forward per distinct net in play: net 0 = central policy, net 1 + k = opponent_nets[k].
        With an LSTM policy the deciding seat's state goes in (multiplied by the previous step's terminal mask, :81,85-89)
        and its new state is kept for the games that really step."""
        N = f.shape[0]
        if not self.opponent_nets:
            groups = [(None, self.policy if self._shadow is None else self._shadow)]
        else:
            ar = torch.arange(N, device=f.device)
            net_id = torch.where(pol == 0, torch.zeros_like(pol), 1 + self.opp_index[ar, (pol - 1).clamp(min=0)])
            # the rows of every net in play, ascending within a net: one stable sort and ONE host read (the group sizes) per pass
            # (round 5: torch.unique(...).tolist() and a nonzero() per net - five host waits per pass)
            order = torch.argsort(net_id, stable=True)
            counts = torch.bincount(net_id, minlength=len(self.opponent_nets) + 1).tolist()
            groups, o = [], 0
            for k, c in enumerate(counts):
                if c:
                    groups.append((order[o:o + c], (self.policy if self._shadow is None else self._shadow) if k == 0 else self.opponent_nets[k - 1]))
                o += c
        actions = torch.zeros((N, spec.ACTION_WORDS), dtype=torch.int64, device=f.device)
        logp = torch.zeros((N,), dtype=torch.float32, device=f.device)
        if self.recurrent:
            ar_all = torch.arange(N, device=f.device)
            seat = deciding - 1
            h_in, c_in = self.hid[0, ar_all, seat], self.hid[1, ar_all, seat]
            new_h, new_c = h_in.clone(), c_in.clone()
        for idx, net in groups:
            args = (f, lists, lens, masks) if idx is None else (f[idx], lists[idx], lens[idx], masks[idx])
            kw = {"generator": self.sample_gen}
            if games is not None and getattr(net, "wants_games", False):
                kw["games"] = games                      # (test policies keyed by game: row j of this pass is game games[j])
            if self.recurrent:
                sel = slice(None) if idx is None else idx
                kw.update(hidden=(h_in[sel], c_in[sel]), nonterminal=term[sel])
            if idx is None and getattr(self, "graph_act", False) and not self.recurrent:
                if self._graphed is None or self._graphed.policy is not net:
                    from .forward_search import GraphedAct
                    self._graphed = GraphedAct(net, buckets=self._bucket_list(), autocast_dtype=self.autocast_dtype, generator=self.sample_gen)
                res = self._graphed(f, lists, lens, masks, with_logp=True, clone=False)
            elif (idx is not None and getattr(self, "graph_act", False) and not self.recurrent and not getattr(net, "wants_games", False)
                  and hasattr(net, "refresh_kernel_packs")):
                # league opponents (round 6): one captured pass per net in play and row-count bucket instead of an eager pass per net
                # (~150 launches each, host-bound: 9.4 s per rollout of T = 200 at 65 536 games against 2.4 s for self-play);
                # the rows beyond the group are padding.  Same generator, registered with every graph.
                from .forward_search import GraphedAct
                g = self._graphed_nets.get(id(net))
                if g is None or g.policy is not net:
                    g = self._graphed_nets[id(net)] = GraphedAct(net, buckets=self._group_buckets(), autocast_dtype=self.autocast_dtype, generator=self.sample_gen)
                res = g(*args, with_logp=True, clone=False)
            elif self.autocast_dtype is not None:
                with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                    res = net.act(*args, **kw)
            else:
                res = net.act(*args, **kw)
            a, lp = res[1], res[2]
            if self.recurrent:
                new_h[sel], new_c[sel] = res[3][0].float(), res[3][1].float()
            if idx is None:
                actions, logp = a, lp[:, 0]
            else:
                actions[idx] = a
                logp[idx] = lp[:, 0]
        if self.recurrent:
            keep = live[:, None]
            self.hid[0, ar_all, seat] = torch.where(keep, new_h, h_in)                     # :89
            self.hid[1, ar_all, seat] = torch.where(keep, new_c, c_in)
        return actions, logp

    def close(self):
        """Drops what the collector holds on the device - the captured policy passes (hipGraphs and their pools), the acting copy of
        the net and the rollout storage - without waiting for the garbage collector; the collector cannot be used afterwards."""
        if self._graphed is not None:
            self._graphed.graphs.clear()
        self._graphed = self._shadow = self.storage = None
        for g in getattr(self, "_graphed_nets", {}).values():
            g.graphs.clear()
        self._graphed_nets = {}
        self.opponent_nets = []

    # game_manager.py:142-150
    def after_rollouts(self):
        st, T, N = self.storage, self.T, self.N
        ar = torch.arange(N, device=self.device)
        last_t = (self.n_obs - 1).clamp(min=0)
        st.obs_f[0] = st.obs_f[last_t, ar]
        st.lists[0] = st.lists[last_t, ar]
        st.lens[0] = st.lens[last_t, ar]
        st.masks[0] = st.masks[(self.n_msk - 1).clamp(min=0, max=T + 1), ar]
        if self.recurrent:
            st.hidden[:, 0] = st.hidden[:, last_t, ar]
        had_obs = self.n_obs > 0
        self.n_obs.copy_(had_obs.long())
        self.n_msk.fill_(1)
        self.n_act.zero_()
        self.n_rew.zero_()
