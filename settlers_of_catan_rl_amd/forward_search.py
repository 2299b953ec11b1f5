"""Batched forward search (reference RL/forward_search_policy/*; BASELINE.json configs[4]).

The reference planner decides ONE move of ONE game: it proposes <= 10 root actions with the policy
(`default_sample_actions`, sample_actions_fn.py:55-328), then for a wall-clock budget lets 11 CPU worker processes run
simulations - restore the root state, `randomise_uncertainty`, play the proposed action, then up to `max_depth` further
decisions of the searching player with the policy acting for every seat, and score the line with a GAE-style estimate
(worker.py:39-143) - allocating simulations to root actions with a UCB rule (policy.py:151-177).

Here R root games are searched at once.  Every piece keeps the reference's arithmetic and quirks:
  * `propose_actions`  - default_sample_actions for all roots: one batched, type-conditioned policy call per step of the
                         procedure (13 first-pass types + <= max_actions refinement draws) instead of one call per root.
  * `simulate`         - run_simulation_forward for n simulations in lock-step on a VecCatanEnv (dense rewards, no
                         auto-reset): incl. the simulator's get_players_turn that ignores the discard phase and the
                         `elif done` reward rule.
  * `gae_estimate`     - worker.gae, vectorised over simulations (the unused final value, the `len(values) <= 1` shortcut).
  * `UCBStats`         - _select_action / _update_stats / MovingAvgCalculator, vectorised over roots (host, float64).
  * `ForwardSearch`    - the act() loop in rounds of K simulations per root (inside a round simulations are allocated one
                         after the other exactly as the reference hands them to free workers), with a fixed simulation
                         budget (the reference's own "deterministic testing" variant, policy.py:112) or the reference's
                         wall-clock budget per root (`max_thinking_time`, policy.py:91,111); LSTM policies carry every
                         seat's (h, c) through proposals and simulations (policy.py:72-106, worker.py:61-95).
State broadcast = catan_state_export / catan_state_import; every simulation gets its own Philox substream.
"""
import math
import random as _py_random

import numpy as np
import torch

from . import spec

MAX_PROP_TRADE_ACTIONS = 3                                   # sample_actions_fn.py:6
PRIORITIES = [["settlement", "city", "move_robber", "steal", "discard"], ["road"], ["play_dev"], ["exchange_res", "prop_trade"]]
TYPE_TO_IND = {"settlement": 0, "road": 1, "city": 2, "buy_dev": 3, "play_dev": 4, "exchange_res": 5, "prop_trade": 6,
               "respond_trade": 7, "move_robber": 8, "roll_dice": 9, "end_turn": 10, "steal": 11, "discard": 12}
MO = spec.MASK_OFFSETS


# ---------------------------------------------------------------------------------------------------- worker.gae
def gae_estimate(values, n_values, rewards, n_rewards, gamma, done, gae_lambda=0.95):
    """worker.py:117-143 for a batch: values [n, *] / rewards [n, *] hold n_values[i] / n_rewards[i] valid entries.
    -> float64 [n]."""
    values = np.asarray(values, dtype=np.float64); rewards = np.asarray(rewards, dtype=np.float64)
    n_values = np.asarray(n_values); n_rewards = np.asarray(n_rewards); done = np.asarray(done, dtype=bool)
    n = values.shape[0]
    out = np.zeros(n, dtype=np.float64)
    short = n_values <= 1                                     # `return rewards[0] + gamma * rewards[1]`
    if short.any():
        assert (n_rewards[short] >= 2).all(), "worker.gae indexes rewards[1] here (IndexError in the reference)"
        out[short] = rewards[short, 0] + gamma * rewards[short, 1]
    idx = np.flatnonzero(~short)
    if idx.size:
        v = values[idx]; r = rewards[idx]; nv = n_values[idx]; d = done[idx]
        num_steps = nv - 2                                    # len(values[:-1]) - 1
        g = np.zeros(idx.size, dtype=np.float64)
        for step in range(int(num_steps.max()) - 1, -1, -1):
            live = step < num_steps
            delta = r[:, 1 + step] + gamma * v[:, step + 1] - v[:, step]      # rewards_for_gae = rewards[1:-1]
            last = live & (step == num_steps - 1) & d
            g = np.where(last, delta, np.where(live, delta + gamma * gae_lambda * g, g))
        out[idx] = r[:, 0] + gamma * (g + v[:, 0])
    return out


# ---------------------------------------------------------------------------------------------------- UCB bookkeeping
class UCBStats(object):
    """policy.py:151-177 + utils.MovingAvgCalculator for R roots; root i has n_actions[i] proposed actions."""

    def __init__(self, n_roots, max_actions, window=500):
        R, A = n_roots, max_actions
        self.R, self.A = R, A
        self.window = np.zeros((R, window)); self.window_size = window
        self.num_added = np.zeros(R, dtype=np.int64)
        self.avg = np.zeros(R); self.var = np.zeros(R); self.last_std = np.zeros(R)
        self.new_decision(np.zeros(R, dtype=np.int64))

    def new_decision(self, n_actions):
        """act(): the per-decision counters restart, the value moving average persists (policy.py:48,96-100)."""
        self.n_actions = np.asarray(n_actions)
        self.finished = np.zeros(self.R, dtype=np.int64)
        self.finished_each = np.zeros((self.R, self.A)); self.started_each = np.zeros((self.R, self.A))
        self.exploit = np.zeros((self.R, self.A))

    def select(self, explore=True):
        """_select_action for every root (first best index wins ties, `score > best_score`)."""
        exploit = self.exploit / (self.finished_each + 1e-5)
        score = exploit
        if explore:
            explore_score = np.sqrt(2.0 * np.log(self.finished + 2)[:, None] / (self.finished_each + self.started_each + 1e-10))
            score = exploit + 2.0 * np.maximum(self.last_std, 1.0)[:, None] * explore_score
        score = np.where(np.arange(self.A)[None, :] < self.n_actions[:, None], score, -np.inf)
        return np.argmax(score, axis=1)

    def start(self, action_id, sel=None):
        """action_id: one id per root (all R); sel: the roots that really start the simulation"""
        rows = np.arange(self.R) if sel is None else np.asarray(sel)
        self.started_each[rows, np.asarray(action_id)[rows] if sel is not None else action_id] += 1

    def update(self, val, action_id, sel=None):
        """_update_stats + MovingAvgCalculator.update for one finished simulation per selected root."""
        rows = np.arange(self.R) if sel is None else np.asarray(sel)
        val = np.asarray(val, dtype=np.float64)
        self.started_each[rows, action_id] -= 1
        self.finished[rows] += 1
        self.finished_each[rows, action_id] += 1
        self.exploit[rows, action_id] += val
        # utils.py:14-46
        k = self.num_added[rows] % self.window_size
        old_value = self.window[rows, k]
        self.window[rows, k] = val
        self.num_added[rows] += 1
        na = self.num_added[rows]
        old_avg = self.avg[rows].copy()
        filling = na <= self.window_size
        delta = np.where(filling, val - old_avg, val - old_value)
        avg = old_avg + np.where(filling, delta / na, delta / self.window_size)
        var = self.var[rows] + np.where(filling, delta * (val - avg), delta * ((val - avg) + (old_value - old_avg)))
        self.avg[rows] = avg; self.var[rows] = var
        variance = np.where(filling, np.where(na == 1, 1.0, var / np.maximum(na - 1, 1)), var / self.window_size)
        with np.errstate(invalid="ignore"):
            std = np.sqrt(variance)
        self.last_std[rows] = np.where(np.isnan(std), 0.1, std)       # `except: std = 0.1` / isnan


# ---------------------------------------------------------------------------------------------------- small-batch inference
class GraphedAct(object):
    """policy.act for SMALL batches as hipGraph replays.  Towards the end of a round only a few simulations are still
    running, and a policy pass is then ~1 300 tiny kernels - launch-bound at ~10 ms whatever the batch.  For a few bucket
    sizes the pass is captured once (torch.cuda.CUDAGraph: the library GEMMs and the hand-written attention / LayerNorm
    launches alike, all on the capture stream) with static input buffers and replayed; rows beyond the live ones are
    padding and ignored.  Sampling uses torch's default CUDA generator, which graphs advance correctly.  Falls back to the
    eager call if capture is unavailable."""

    def __init__(self, policy, buckets=(512, 4096, 16384), autocast_dtype=None, deterministic=False, generator=None):
        """generator: the CUDA torch.Generator the sampling draws from (registered with every captured graph, so replays
        advance it as eager calls would); None = torch's default CUDA generator."""
        self.policy, self.buckets, self.autocast_dtype, self.deterministic = policy, tuple(sorted(buckets)), autocast_dtype, deterministic
        self.generator = generator
        self.graphs = {}
        self.failed = False

    def _run(self, f, lists, lens, masks):
        kw = {} if self.generator is None else {"generator": self.generator}
        if self.autocast_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                return self.policy.act(f, lists, lens, masks, deterministic=self.deterministic, **kw)
        return self.policy.act(f, lists, lens, masks, deterministic=self.deterministic, **kw)

    def _capture(self, B, f, lists, lens, masks):
        st = {"f": f[:1].expand(B, -1).clone(), "lists": lists[:1].expand(B, -1, -1).clone(),
              "lens": lens[:1].expand(B, -1).clone(), "masks": masks[:1].expand(B, -1).clone()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._run(st["f"], st["lists"], st["lens"], st["masks"])
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        if self.generator is not None:
            g.register_generator_state(self.generator)
        with torch.cuda.graph(g):
            v, a, lp = self._run(st["f"], st["lists"], st["lens"], st["masks"])[:3]
        st["g"], st["v"], st["a"], st["lp"] = g, v, a, lp
        st["sig"] = self._signature()
        return st

    def _signature(self):
        """where the policy's parameters live: a captured graph reads exactly these addresses"""
        ps = list(self.policy.parameters())
        return (len(ps), hash(tuple(p.data_ptr() for p in ps)), ps[0].dtype) if ps else ()   # EVERY parameter: one replaced in the middle is stale too

    def static_inputs(self, B):
        """The captured graph's own input buffers (f, lists, lens, masks) for bucket B, or None before the capture: a producer that
        writes straight into them (the collector's k_obs_rows / mask expansion) saves the copy of every replay."""
        st = self.graphs.get(B)
        return None if st is None else (st["f"], st["lists"], st["lens"], st["masks"])

    def __call__(self, f, lists, lens, masks, with_logp=False, clone=True):
        """-> (value [n,1], actions [n,18]) (+ log-prob [n,1] with `with_logp`) for n rows; eager when n exceeds the largest
        bucket.  Inputs that ARE the graph's static buffers (static_inputs) are not copied; clone=False hands out the graph's
        output buffers themselves (valid until the next replay)."""
        n = f.shape[0]
        B = next((b for b in self.buckets if n <= b), None)
        if B is None or self.failed or not f.is_cuda:
            v, a, lp = self._run(f, lists, lens, masks)[:3]
            return (v, a, lp) if with_logp else (v, a)
        if B not in self.graphs:
            try:
                self.graphs[B] = self._capture(B, f, lists, lens, masks)
            except Exception:                                   # capture not available: stay eager
                self.failed = True
                torch.cuda.synchronize()
                v, a, lp = self._run(f, lists, lens, masks)[:3]
                return (v, a, lp) if with_logp else (v, a)
        st = self.graphs[B]
        if st["sig"] != self._signature():                  # the parameters moved (a .to() / a rebuilt module): the graph is stale
            self.graphs.clear()
            return self.__call__(f, lists, lens, masks, with_logp)
        refresh = getattr(self.policy, "refresh_kernel_packs", None)
        if refresh is not None:
            refresh()                                           # host-side parameter packs a replay would not rebuild
        for k, x in (("f", f), ("lists", lists), ("lens", lens), ("masks", masks)):
            if x.data_ptr() != st[k].data_ptr():
                st[k][:n] = x
        st["g"].replay()
        out = (st["v"][:n], st["a"][:n], st["lp"][:n]) if with_logp else (st["v"][:n], st["a"][:n])
        return tuple(o.clone() for o in out) if clone else out


# ---------------------------------------------------------------------------------------------------- simulations
@torch.no_grad()
def simulate(env, policy, ctrl, init_actions, max_depth=20, gamma=0.999, deterministic=False, generator=None, autocast_dtype=None,
             graphed=None, hidden=None, init_hidden=None):
    """run_simulation_forward (worker.py:61-114) for all env.n simulations in lock-step.  env: dense-reward, no auto-reset,
    already holding the (randomised) start states; ctrl int [n] PlayerId of the searching player; init_actions [n,18].
    LSTM policies (`include_lstm`): hidden [2, n, 4, L] = every seat's (h, c) at the start state (`curr_hidden_states`),
    init_hidden [2, n, L] = the searching seat's state after it took the initial action (`init_player_hs`, worker.py:73); each
    decision runs from the deciding seat's state and stores the new one (:84-95; the terminal mask is 1 throughout).
    -> float64 numpy [n] value estimates."""
    n, dev = env.n, env.device
    ar = torch.arange(n, device=dev)
    ctrl = torch.as_tensor(ctrl, device=dev).long()
    D = max_depth
    rewards = torch.zeros((n, D + 2), dtype=torch.float64, device=dev); n_rew = torch.zeros(n, dtype=torch.long, device=dev)
    values = torch.zeros((n, D + 1), dtype=torch.float64, device=dev); n_val = torch.zeros(n, dtype=torch.long, device=dev)

    def push(buf, cnt, sel, x):                         # append x[i] to row i's list where sel[i] (masked write: no host read)
        c = cnt.clamp(max=buf.shape[1] - 1)
        buf[ar, c] = torch.where(sel, x.double(), buf[ar, c])
        cnt += sel.long()

    reward, done = env.step(torch.as_tensor(init_actions, device=dev).to(torch.int32))      # worker.py:71
    reward = reward.clone(); first_done = done.bool().clone()
    push(rewards, n_rew, torch.ones(n, dtype=torch.bool, device=dev), reward[ar, ctrl - 1])
    agent_actions = torch.ones(n, dtype=torch.long, device=dev)
    active = ~first_done                                                                    # `if done: return actual_rewards[0]`
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    rec = bool(getattr(policy, "include_lstm", False))
    if rec:
        L = int(policy.lstm_size)
        hid = torch.zeros((2, n, 4, L), dtype=torch.float32, device=dev) if hidden is None else hidden.to(dev).float().clone()
        if init_hidden is not None:
            hid[:, ar, ctrl - 1] = init_hidden.to(dev).float()                              # worker.py:73
        graphed = None
    # (device env with the game-list views: see the loop)
    lists_of_games = (not rec) and hasattr(env, "get_obs_rows") and hasattr(env, "L") and getattr(env, "device", None) is not None and torch.device(env.device).type == "cuda"
    while True:
        live = active & (agent_actions < D)                                                 # worker.py:81
        if not bool(live.any()):
            break
        players_go = env.players_turn_sim().long()                                          # :82, 146-151
        # only the simulations still running need a decision (the batch thins out as lines reach max_depth or end)
        idx = live.nonzero(as_tuple=True)[0]
        sub = idx.numel() < n
        if lists_of_games and graphed is not None and idx.numel() <= graphed.buckets[-1]:
            # the observations and masks of the LIVE games only, straight into the captured pass's input buffers (catan_obs_rows_of /
            # catan_masks_of; bf16 under bf16 autocast: every value is a multiple of 1/8): all games' fp32 observations + a gather of
            # the live rows + the net's cast + the copy into the graph's buffers were 15 % of a batch of root decisions
            games = idx.to(torch.int32)
            nl = int(idx.numel())
            B = next(b for b in graphed.buckets if nl <= b)
            bufs = graphed.static_inputs(B)
            odt = autocast_dtype if autocast_dtype in (torch.bfloat16,) else torch.float32
            if bufs is not None and bufs[0].dtype == odt and bufs[1].dtype == torch.int32 and bufs[2].dtype == torch.int32 and bufs[3].dtype == torch.float32:
                f, lists, lens = env.get_obs_rows(odt, out=tuple(x[:nl] for x in bufs[:3]), games=games)
                masks = env.get_action_masks(bufs[3][:nl], games=games)
            else:
                f, lists, lens = env.get_obs_rows(odt, games=games)
                masks = env.get_action_masks(games=games)
            args = (f, lists, lens, masks)
        else:
            f, lists, lens = env.get_obs()
            masks = env.get_action_masks()
            args = (f[idx], lists[idx], lens[idx].long(), masks[idx]) if sub else (f, lists, lens.long(), masks)
        kw = {}
        if rec:
            seat = players_go[idx] - 1
            kw = {"hidden": (hid[0, idx, seat], hid[1, idx, seat]), "nonterminal": torch.ones(idx.numel(), device=dev)}
        if graphed is not None and idx.numel() <= graphed.buckets[-1]:
            value_s, action_s = graphed(*args)                                             # hipGraph replay of the pass at the next bucket size (a pass is launch-bound at every width up to 65 536 rows)
        elif autocast_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=autocast_dtype):
                res = policy.act(*args, deterministic=deterministic, generator=generator, **kw)
            value_s, action_s = res[0], res[1]
        else:
            res = policy.act(*args, deterministic=deterministic, generator=generator, **kw)
            value_s, action_s = res[0], res[1]
        if rec:
            hid[0, idx, seat], hid[1, idx, seat] = res[3][0].float(), res[3][1].float()     # :95
        if sub:
            value = torch.zeros((n, 1), dtype=value_s.dtype, device=dev); value[idx] = value_s
            action = torch.zeros((n, spec.ACTION_WORDS), dtype=action_s.dtype, device=dev); action[idx] = action_s
        else:
            value, action = value_s, action_s
        value = policy.denormalise(value.float()[:, 0])                                     # :92-93
        a_env = action.to(torch.int32)
        a_env[:, 0] = torch.where(live, a_env[:, 0], torch.full_like(a_env[:, 0], -1))     # finished simulations idle
        reward, done = env.step(a_env)
        done = done.bool() & live
        r_ctrl = reward[ar, ctrl - 1]
        is_agent = live & (players_go == ctrl)                                              # :99-104
        agent_actions += is_agent.long()
        push(rewards, n_rew, is_agent | (live & ~is_agent & done), r_ctrl)
        push(values, n_val, is_agent, value)
        finished |= done                                                                    # :109-111
        active &= ~done
    fd = first_done.cpu().numpy()
    rewards_h = rewards.cpu().numpy()
    out = rewards_h[:, 0].copy()                                                            # `if done: return actual_rewards[0]`
    rest = np.flatnonzero(~fd)
    if rest.size:
        out[rest] = gae_estimate(values.cpu().numpy()[rest], n_val.cpu().numpy()[rest], rewards_h[rest], n_rew.cpu().numpy()[rest],
                                 gamma, finished.cpu().numpy()[rest])
    return out


# ---------------------------------------------------------------------------------------------------- proposals
def _update_action_masks(action, m):
    """sample_actions_fn.py:31-53 on one flat mask row (numpy float [325]) - removes the proposed target from its head."""
    t = int(action[0])
    if t == 0:
        if m[MO[1]:MO[1] + 54].sum() > 1: m[MO[1] + action[1]] = 0
    elif t == 1:
        if m[MO[2]:MO[2] + 73].sum() > 1: m[MO[2] + action[2]] = 0
    elif t == 2:
        if m[MO[1] + 54:MO[1] + 108].sum() > 1: m[MO[1] + 54 + action[1]] = 0
    elif t == 4:
        if m[MO[4]:MO[4] + 5].sum() > 1: m[MO[4] + action[4]] = 0
    elif t == 8:
        if m[MO[3]:MO[3] + 19].sum() > 0: m[MO[3] + action[3]] = 0
    elif t == 11:
        if m[MO[6] + 3:MO[6] + 6].sum() > 0: m[MO[6] + 3 + action[6]] = 0
    elif t == 12:
        if m[MO[11]:MO[11] + 5].sum() > 1: m[MO[11] + action[17]] = 0


@torch.no_grad()
def propose_actions(policy, f, lists, lens, masks, max_actions=10, initial_settlement_phase=None,
                    consider_all_initial_settlements=False, rngs=None, deterministic=False, generator=None, autocast_dtype=None,
                    hidden=None, return_hidden=False):
    """default_sample_actions (sample_actions_fn.py:55-328, with dont_propose_devcards = dont_propose_trades = False) for R
    roots.  f/lists/lens/masks: the roots' observations and masks (device tensors); rngs: one `random.Random` per root for
    the procedure's random.choice calls.  hidden (LSTM policies): (h, c) [R, L] each, the deciding seat's state; every proposal
    is sampled from it, and the state after the decision - the same for all proposals of a root, the LSTM step precedes the
    action heads - is returned with return_hidden (`player_next_hidden_states`, sample_actions_fn.py).
    -> (actions int64 numpy [R, A, 18], counts [R]) in proposal order [, (h, c) [R, L]]."""
    R, dev = f.shape[0], f.device
    m = masks.detach().cpu().numpy().astype(np.float32).copy()            # working copies, updated as targets are used up
    type_masks = m[:, MO[0]:MO[0] + 13].copy()
    rngs = rngs or [_py_random.Random(i) for i in range(R)]
    init_phase = np.zeros(R, dtype=bool) if initial_settlement_phase is None else np.asarray(initial_settlement_phase, dtype=bool)
    effective = np.zeros(R)
    trades = np.zeros(R, dtype=np.int64)

    rec = bool(getattr(policy, "include_lstm", False))
    if rec:
        Lh = int(policy.lstm_size)
        h_in = (torch.zeros(R, Lh, device=dev), torch.zeros(R, Lh, device=dev)) if hidden is None else (hidden[0].to(dev).float(), hidden[1].to(dev).float())
        h_next = (h_in[0].clone(), h_in[1].clone())

    def act(rows, types):
        idx = torch.as_tensor(rows, device=dev, dtype=torch.long)
        forced = torch.as_tensor(types, device=dev, dtype=torch.long)
        mk = torch.from_numpy(m[rows]).to(dev)
        args = (f[idx], lists[idx], lens[idx].long(), mk)
        kw = {"hidden": (h_in[0][idx], h_in[1][idx]), "nonterminal": torch.ones(idx.numel(), device=dev)} if rec else {}
        if autocast_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=autocast_dtype):
                res = policy.act(*args, deterministic=deterministic, generator=generator, condition_on_action_type=forced, **kw)
        else:
            res = policy.act(*args, deterministic=deterministic, generator=generator, condition_on_action_type=forced, **kw)
        if rec:
            h_next[0][idx], h_next[1][idx] = res[3][0].float(), res[3][1].float()
        return res[1].cpu().numpy()

    # Bookkeeping as arrays over the roots (round 1 walked Python dicts / lists per root: 1.5 s at 4 096 roots):
    #   avail [R, 13]   refinement draws left per type (sample_actions_fn.py `available_actions`; only refinable types > 0)
    #   props [R, cap, 18], n_prop [R]   the proposals in order;   ex [R, cap, 2], n_ex [R]   exchange pairs proposed so far
    REFINABLE = (0, 1, 2, 4, 5, 6, 8, 11, 12)
    cap = max(int(max_actions), 13) + (54 if consider_all_initial_settlements else 0) + 1
    avail = np.zeros((R, 13))
    props = np.zeros((R, cap, spec.ACTION_WORDS), dtype=np.int64); n_prop = np.zeros(R, dtype=np.int64)
    ex = np.zeros((R, cap, 2), dtype=np.int64); n_ex = np.zeros(R, dtype=np.int64)

    def counts_for(t, rows):
        """`count - 1` further targets of type t (sample_actions_fn.py:117-196), for all `rows` at once"""
        mr = m[rows]
        if t == 0: return mr[:, MO[1]:MO[1] + 54].sum(1) - 1
        if t == 1: return mr[:, MO[2]:MO[2] + 73].sum(1) - 1
        if t == 2: return mr[:, MO[1] + 54:MO[1] + 108].sum(1) - 1
        if t == 4: return mr[:, MO[4]:MO[4] + 5].sum(1) - 1
        if t == 5: return mr[:, MO[10]:MO[10] + 5].sum(1) * mr[:, MO[9]:MO[9] + 5].sum(1) - 1
        if t == 6: return np.full(len(rows), MAX_PROP_TRADE_ACTIONS - 1.0)
        if t == 8: return mr[:, MO[3]:MO[3] + 19].sum(1) - 1
        if t == 11: return mr[:, MO[6] + 3:MO[6] + 6].sum(1) - 1
        return mr[:, MO[11]:MO[11] + 5].sum(1) - 1                        # 12

    # _update_action_masks (sample_actions_fn.py:31-53) for many roots at once: (first mask column, width, action word, minimum sum)
    TARGET = {0: (MO[1], 54, 1, 1), 1: (MO[2], 73, 2, 1), 2: (MO[1] + 54, 54, 1, 1), 4: (MO[4], 5, 4, 1), 8: (MO[3], 19, 3, 0),
              11: (MO[6] + 3, 3, 6, 0), 12: (MO[11], 5, 17, 1)}

    def use_up(t, rows, acts):
        if t not in TARGET or len(rows) == 0:
            return
        lo, w, word, least = TARGET[t]
        rows = np.asarray(rows)
        ok = m[rows, lo:lo + w].sum(1) > least
        m[rows[ok], lo + acts[ok, word]] = 0

    def append(rows, acts):
        rows = np.asarray(rows)
        props[rows, n_prop[rows]] = acts
        n_prop[rows] += 1

    name_of = {v: k for k, v in TYPE_TO_IND.items()}
    for t in range(13):                                                   # first pass: one proposal per available type
        rows = np.flatnonzero(type_masks[:, t] == 1)
        if rows.size == 0:
            continue
        if t in REFINABLE:
            c = counts_for(t, rows).astype(np.float64)
            avail[rows, t] = c
            effective[rows] += c
        elif t == 7:
            effective[rows] += m[rows, MO[5]:MO[5] + 2].sum(1) - 1         # respond: counted but not refinable (:198)
        a = act(rows, np.full(rows.size, t))
        assert (a[:, 0] == t).all()
        append(rows, a)
        if t == 5:
            ex[rows, n_ex[rows]] = a[:, 15:17]; n_ex[rows] += 1
        if t == 6:
            trades[rows] += 1
        use_up(t, rows, a)
    # refinement draws (:286-328)
    if consider_all_initial_settlements:
        n_more = np.where(init_phase, avail[:, 0].astype(np.int64), np.minimum(max_actions - n_prop, effective).astype(np.int64))
    else:
        n_more = np.minimum(max_actions - n_prop, effective).astype(np.int64)
    alive = n_more > 0
    prio = [[TYPE_TO_IND[x] for x in pr] for pr in PRIORITIES]
    for i in range(int(n_more.max()) if R else 0):
        rows, types = [], []
        for r in np.flatnonzero(alive & (i < n_more)):
            ac_type = -1
            av_r = avail[r]
            for pr in prio:
                av = [x for x in pr if av_r[x] > 0]
                if trades[r] >= MAX_PROP_TRADE_ACTIONS and 6 in av:
                    av.remove(6)
                if av:
                    ac_type = TYPE_TO_IND[rngs[r].choice([name_of[x] for x in av])]
                    av_r[ac_type] -= 1
                    break
            if ac_type < 0:
                alive[r] = False                                          # "something gone wrong - just return what we have"
                continue
            rows.append(r); types.append(ac_type)
        if not rows:
            break
        rows, types = np.asarray(rows), np.asarray(types)
        a = act(rows, types)
        is_ex = types == 5
        for j in np.flatnonzero(is_ex):
            # the reference re-draws the pair `while prop_exchange not in exchanges_proposed` (:316-320): a NEW pair is
            # replaced by random legal picks until it coincides with one proposed before
            r, act_j = rows[j], a[j]
            seen = {(int(x), int(y)) for x, y in ex[r, :n_ex[r]]}
            pair = (int(act_j[15]), int(act_j[16]))
            give_ok = np.flatnonzero(m[r, MO[9]:MO[9] + 5]); recv_ok = np.flatnonzero(m[r, MO[10]:MO[10] + 5])
            while pair not in seen:
                act_j[15] = rngs[r].choice(list(give_ok)); act_j[16] = rngs[r].choice(list(recv_ok))
                pair = (int(act_j[15]), int(act_j[16]))
            ex[r, n_ex[r]] = pair; n_ex[r] += 1
        append(rows, a)
        trades[rows[types == 6]] += 1
        for t in TARGET:
            sel = types == t
            if sel.any():
                use_up(t, rows[sel], a[sel])
    A = int(n_prop.max()) if R else 0
    out = props[:, :A].copy()
    counts = n_prop.copy()
    if return_hidden:
        return out, counts, (h_next if rec else None)
    return out, counts


# ---------------------------------------------------------------------------------------------------- the planner
class ForwardSearch(object):
    """ForwardSearchPolicy.act (policy.py:72-149) for all games of `root_env` at once, each searched by its deciding
    player.  make_sim_env(n) -> an env of n games with dense rewards and no auto-reset (VecCatanEnv on the GPU)."""

    def __init__(self, policy, make_sim_env, n_roots, max_init_actions=10, max_depth=20, gamma=0.999, sims_per_root=64,
                 sims_per_round=16, consider_all_moves_for_opening_placement=False, seed=0, autocast_dtype=None, use_graphs=False):
        assert sims_per_root % sims_per_round == 0
        if autocast_dtype is not None and hasattr(policy, "inference_copy") and getattr(policy, "_inference_dtype", None) is None:
            policy = policy.inference_copy(autocast_dtype)      # weights in the autocast dtype: no per-call casts
        self.policy, self.R = policy, n_roots
        self.max_init_actions, self.max_depth, self.gamma = max_init_actions, max_depth, gamma
        self.S, self.K = sims_per_root, sims_per_round
        self.consider_all = consider_all_moves_for_opening_placement
        self.sim_env = make_sim_env(n_roots * sims_per_round)
        self.stats = UCBStats(n_roots, 54 if consider_all_moves_for_opening_placement else max_init_actions)
        self.rngs = [_py_random.Random(seed * 1000003 + i) for i in range(n_roots)]
        self.gen = None
        self.autocast_dtype = autocast_dtype
        if torch.cuda.is_available() and next(policy.parameters()).is_cuda:
            from . import nn_kernels
            nn_kernels.use_tuned_gemms()
        self.graphed = GraphedAct(policy, buckets=(512, 4096, 16384, 32768, 49152, 65536), autocast_dtype=autocast_dtype) if use_graphs else None
        self.sims_run = 0
        self._rng_word = spec.STATE_OFFSETS["rng_draws"][0]

    @torch.no_grad()
    def act(self, root_env, initial_settlement=None, deterministic=False, hidden=None, zero_opponent_hidden_states=False,
            max_thinking_time=None):
        """-> (actions int64 numpy [R,18], info dict).  Roots with a single proposal skip the search (policy.py:88-89).
        LSTM policies: hidden [2, R, 4, L] = the (h, c) of every seat of every root (`curr_hidden_states`, policy.py:72; zeros if
        omitted); proposals and simulations run from them as the reference's do (:73-86, worker.py:61-95), and
        info["next_hidden"] [2, R, L] is the searching seat's state after its decision (`player_next_hidden_states`).
        max_thinking_time (seconds; None = the fixed budget of `sims_per_root`): the reference's wall-clock budget
        (policy.py:91,112-138): root r thinks for max_thinking_time * n_proposed[r] / max_init_actions; rounds of K simulations
        per root are run until the longest of those budgets is spent, and a root stops taking results once its own is."""
        R, K = self.R, self.K
        assert root_env.n == R
        ctrl = root_env.deciding_player().long()
        f, lists, lens = root_env.get_obs()
        masks = root_env.get_action_masks()
        rec = bool(getattr(self.policy, "include_lstm", False))
        hid_sim = init_sim = next_hidden = None
        if rec:
            Lh = int(self.policy.lstm_size)
            hidden = torch.zeros((2, R, 4, Lh), device=root_env.device) if hidden is None else hidden.to(root_env.device).float()
            ar_r = torch.arange(R, device=root_env.device)
            own = (hidden[0, ar_r, ctrl - 1], hidden[1, ar_r, ctrl - 1])
        props, counts, nh = propose_actions(self.policy, f, lists, lens, masks, self.max_init_actions, initial_settlement, self.consider_all,
                                            self.rngs, deterministic, self.gen, self.autocast_dtype, hidden=own if rec else None, return_hidden=True)
        if rec:
            next_hidden = torch.stack(nh)                                           # [2, R, L]
            if zero_opponent_hidden_states:                                         # policy.py:81-86
                keep = torch.zeros((R, 4), dtype=torch.bool, device=root_env.device)
                keep[ar_r, ctrl - 1] = True
                hidden = hidden * keep[None, :, :, None]
            hid_sim = hidden.repeat_interleave(K, dim=1)
            init_sim = next_hidden.repeat_interleave(K, dim=1)
        self.stats.new_decision(counts)
        blobs = root_env.export_state()                                             # [R, 736] int32 (state broadcast)
        blobs = blobs.repeat_interleave(K, dim=0).clone()
        ctrl_sim = ctrl.repeat_interleave(K)
        base_draws = blobs[:, self._rng_word].long() & 0xFFFFFFFF
        props_t = torch.from_numpy(props).to(root_env.device)
        ar_sim = torch.arange(R * K, device=root_env.device)
        import time as _time
        t_start = _time.perf_counter()
        think = None if max_thinking_time is None else max_thinking_time * (counts / float(self.max_init_actions))      # policy.py:91
        rnd = -1
        while True:
            rnd += 1
            if think is None:
                if rnd >= self.S // K:
                    break
                thinking = np.ones(R, dtype=bool)
            else:
                thinking = (_time.perf_counter() - t_start) < think                 # `while elapsed_time < thinking_time`
                if not thinking.any():
                    break
            ids = np.zeros((R, K), dtype=np.int64)
            for k in range(K):                                                      # one simulation after the other (:118-125)
                a = self.stats.select(explore=True)
                self.stats.start(a, np.flatnonzero(thinking))
                ids[:, k] = a
            # every simulation its own philox substream: offset the game stream's draw counter by a large stride
            sub = (base_draws + (1 + rnd * K + (ar_sim % K)) * (1 << 22)) & 0xFFFFFFFF
            blobs[:, self._rng_word] = torch.where(sub >= 2 ** 31, sub - 2 ** 32, sub).to(torch.int32)
            self.sim_env.import_state(blobs)                                        # worker.py:44-45
            self.sim_env.randomise_uncertainty(ctrl_sim)                            # :46
            init = props_t[torch.arange(R, device=props_t.device).repeat_interleave(K), torch.from_numpy(ids.reshape(-1)).to(props_t.device)]
            vals = simulate(self.sim_env, self.policy, ctrl_sim, init, self.max_depth, self.gamma, deterministic, self.gen,
                            self.autocast_dtype, None if deterministic else self.graphed, hidden=hid_sim, init_hidden=init_sim).reshape(R, K)
            rows = np.flatnonzero(thinking)
            for k in range(K):
                self.stats.update(vals[rows, k], ids[rows, k], rows)
            self.sims_run += int(rows.size) * K
        best = self.stats.select(explore=False)
        best = np.where(counts == 1, 0, best)
        chosen = props[np.arange(R), best]
        return chosen, {"n_proposed": counts, "best": best, "finished_each": self.stats.finished_each.copy(),
                        "mean_value": self.stats.exploit / np.maximum(self.stats.finished_each, 1), "next_hidden": next_hidden}
