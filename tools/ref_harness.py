"""Drives the imported upstream reference under the build's per-env RNG contract.

Development container only (needs /root/reference).  Used by tools/gen_golden.py and
tools/fuzz_oracle_vs_ref.py.  Nothing here ships to the GPU box except the fixtures it writes.

RNG contract (DESIGN.md "RNG"):
  source(B) = Philox4x32-10, key = (seed_lo, seed_hi), counter = (draw>>2, stream, env_lo, env_hi),
  output word = block[draw & 3]; stream 0 = game draws.
  bounded(max)  = masked rejection:  v = next_u32 & mask(max) until v <= max
                  (same rule as numpy's legacy rk_interval, so shuffle/randint keep their shape)
  shuffle(x)    = for i = n-1 .. 1: j = bounded(i); swap(x[i], x[j])       (np.random.shuffle)
  randint(1,7)  = 1 + bounded(5)                                           (np.random.randint)
  choice(seq)   = seq[bounded(len-1)]                                      (random.choice, steal)
The reference's three call sites (np.random.shuffle, np.random.randint, random.choice; reference
game/components/board.py:72-84, game/game.py:42,77,139-140,643) are monkey-patched to pull from the
current env's stream.
"""
import contextlib
import random as _py_random
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ref_bootstrap import bootstrap  # noqa: E402

bootstrap()
from env.wrapper import EnvWrapper  # noqa: E402  (reference)
from game.enums import PlayerId, Resource, DevelopmentCard, BuildingType  # noqa: E402  (reference)

from settlers_of_catan_rl_amd import spec  # noqa: E402

M32 = 0xFFFFFFFF
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11; Random123).  ctr: 4 u32, key: 2 u32 -> 4 u32."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> 32, p0 & M32
        hi1, lo1 = p1 >> 32, p1 & M32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & M32, lo1, (hi0 ^ c3 ^ k1) & M32, lo0
        k0 = (k0 + PHILOX_W0) & M32
        k1 = (k1 + PHILOX_W1) & M32
    return c0, c1, c2, c3


class PhiloxStream(object):
    def __init__(self, seed, env_id, stream=0):
        self.key = (seed & M32, (seed >> 32) & M32)
        self.env = (env_id & M32, (env_id >> 32) & M32)
        self.stream = stream
        self.draws = 0

    def next_u32(self):
        d = self.draws
        self.draws += 1
        blk = philox4x32_10((d >> 2, self.stream, self.env[0], self.env[1]), self.key)
        return blk[d & 3]

    def bounded(self, mx):
        if mx == 0:
            return 0
        mask = mx
        mask |= mask >> 1
        mask |= mask >> 2
        mask |= mask >> 4
        mask |= mask >> 8
        mask |= mask >> 16
        while True:
            v = self.next_u32() & mask
            if v <= mx:
                return v

    def shuffle(self, x):
        for i in range(len(x) - 1, 0, -1):
            j = self.bounded(i)
            x[i], x[j] = x[j], x[i]

    def randint(self, lo, hi):
        return lo + self.bounded(hi - lo - 1)

    def choice(self, seq):
        return seq[self.bounded(len(seq) - 1)]


_CURRENT = [None]


@contextlib.contextmanager
def patched_rng(stream):
    """Route the reference's RNG call sites to `stream` while inside the context."""
    saved = (np.random.shuffle, np.random.randint, _py_random.choice, _py_random.shuffle)
    prev = _CURRENT[0]
    _CURRENT[0] = stream
    np.random.shuffle = lambda x: stream.shuffle(x)
    np.random.randint = lambda lo, hi=None: stream.randint(lo, hi)
    _py_random.choice = lambda seq: stream.choice(seq)
    _py_random.shuffle = lambda x: stream.shuffle(x)       # only Game.randomise_uncertainty (game.py:1250,1255) uses it
    try:
        yield stream
    finally:
        np.random.shuffle, np.random.randint, _py_random.choice, _py_random.shuffle = saved
        _CURRENT[0] = prev


PIDS = [PlayerId.White, PlayerId.Blue, PlayerId.Orange, PlayerId.Red]  # value order 1..4
RES = [Resource.Brick, Resource.Wood, Resource.Ore, Resource.Sheep, Resource.Wheat]  # value order 1..5
LABELS = ["next", "next_next", "next_next_next"]


class RefEnv(object):
    """One reference EnvWrapper bound to one Philox stream."""

    def __init__(self, seed, env_id, **env_kwargs):
        self.stream = PhiloxStream(seed, env_id)
        scratch = PhiloxStream(seed ^ 0xABCDEF, env_id)
        with patched_rng(scratch):      # constructor draws (two board resets + one game reset) are discarded
            self.env = EnvWrapper(**env_kwargs)
        self.last_obs = None

    def reset(self):
        with patched_rng(self.stream):
            self.last_obs = self.env.reset()
        return self.last_obs

    def clone(self):
        """an independent copy (deep copy of the whole wrapper: the reference's own restore_state leaves e.g. a harbour
        gained since the save in place, game.py:1093-1205, so it cannot serve as an undo)"""
        import copy
        c = RefEnv.__new__(RefEnv)
        c.env = copy.deepcopy(self.env)
        c.stream = PhiloxStream(0, 0)
        c.stream.key, c.stream.env, c.stream.stream, c.stream.draws = self.stream.key, self.stream.env, self.stream.stream, self.stream.draws
        c.last_obs = self.last_obs
        return c

    def masks(self):
        return self.env.get_action_masks()

    def step(self, action18):
        a = action_to_heads(action18)
        with patched_rng(self.stream):
            obs, reward, done, info = self.env.step(a)
        self.last_obs = obs
        self.last_reward64 = np.array([reward[p] for p in PIDS], dtype=np.float64)      # the reference's Python floats
        rew = self.last_reward64.astype(np.float32)                                      # one rounding (process_batch.py:63)
        return obs, rew, bool(done)

    def deciding_player(self):
        g = self.env.game
        if g.players_need_to_discard:
            return int(g.players_to_discard[0])
        if g.must_respond_to_trade:
            return int(g.proposed_trade["target_player"])
        return int(g.players_go)

    def state_blob(self):
        return state_blob(self.env, self.stream.draws)


def action_to_heads(a):
    a = [int(x) for x in a]
    return [a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7:11], a[11:15], a[15], a[16], a[17]]


def masks_flat(masks):
    return np.concatenate([np.asarray(m, dtype=np.float32).reshape(-1) for m in masks])


def obs_flat(obs):
    """-> (float32[1787], int32[5][25] lists zero padded, int32[5] lengths, player_id)"""
    parts = []
    for k in spec.OBS_FLOAT_KEYS:
        v = obs[k]
        if k == "tile_representations":
            v = np.stack([np.asarray(t, dtype=np.float32) for t in v])
        parts.append(np.asarray(v, dtype=np.float32).reshape(-1))
    f = np.concatenate(parts)
    lists = np.zeros((5, spec.OBS_LIST_PAD), dtype=np.int32)
    lens = np.zeros((5,), dtype=np.int32)
    for i, k in enumerate(spec.OBS_LIST_KEYS):
        v = np.asarray(obs[k]).astype(np.int32)
        lens[i] = len(v)
        lists[i, :len(v)] = v
    return f, lists, lens, int(obs["player_id"])


def state_blob(env, rng_draws=0):
    g = env.game
    b = g.board
    out = np.zeros((spec.STATE_WORDS,), dtype=np.int32)

    def put(name, vals):
        off, n = spec.STATE_OFFSETS[name]
        vals = list(vals)
        assert len(vals) == n, (name, len(vals), n)
        out[off:off + n] = vals

    put("tile_res", [int(t.resource) for t in b.tiles])
    put("tile_val", [int(t.value) for t in b.tiles])
    put("robber_tile", [int(b.robber_tile.id)])
    put("harbour_type", [int(h.id) for h in b.harbours])
    put("corner_bld", [0 if c.building is None else (1 if c.building.type == BuildingType.Settlement else 2)
                       for c in b.corners])
    put("corner_owner", [0 if c.building is None else int(c.building.owner) for c in b.corners])
    put("edge_owner", [0 if e.road is None else int(e.road) for e in b.edges])
    for p in PIDS:
        pl = g.players[p]
        pre = f"p{int(p)}_"
        put(pre + "res", [int(pl.resources[r]) for r in RES])
        put(pre + "vis", [int(pl.visible_resources[r]) for r in RES])
        put(pre + "opp_min", [int(pl.opponent_min_res[l][r]) for l in LABELS for r in RES])
        put(pre + "opp_max", [int(pl.opponent_max_res[l][r]) for l in LABELS for r in RES])
        hb = [0] * 6
        for key in pl.harbours:
            hb[0 if key is None else int(key)] = 1
        put(pre + "harbours", hb)
        put(pre + "n_hidden", [len(pl.hidden_cards)])
        put(pre + "hidden", [int(c) for c in pl.hidden_cards] + [-1] * (25 - len(pl.hidden_cards)))
        put(pre + "n_played", [len(pl.visible_cards)])
        put(pre + "played", [int(c) for c in pl.visible_cards] + [-1] * (25 - len(pl.visible_cards)))
        put(pre + "vp", [int(pl.victory_points)])
    put("bank_res", [int(g.resource_bank[r]) for r in RES])
    put("settlements_left", [int(g.building_bank["settlements"][p]) for p in PIDS])
    put("cities_left", [int(g.building_bank["cities"][p]) for p in PIDS])
    pile = [int(c) for c in g.development_cards_pile]
    put("pile_len", [len(pile)])
    put("pile", pile + [-1] * (25 - len(pile)))
    put("player_order", [int(p) for p in g.player_order])
    put("player_order_id", [int(g.player_order_id)])
    put("players_go", [int(g.players_go)])
    put("initial_phase", [int(g.initial_placement_phase)])
    put("init_settlements", [int(g.initial_settlements_placed[p]) for p in PIDS])
    put("init_roads", [int(g.initial_roads_placed[p]) for p in PIDS])
    put("init_second_corner", [-1 if g.initial_second_settlement_corners[p] is None
                               else int(g.initial_second_settlement_corners[p]) for p in PIDS])
    put("dice_rolled", [int(g.dice_rolled_this_turn)])
    put("played_dev", [int(g.played_development_card_this_turn)])
    put("must_use_dev", [int(g.must_use_development_card_ability)])
    put("must_respond", [int(g.must_respond_to_trade)])
    tr = g.proposed_trade
    if tr is not None:
        give = [int(r) for r in tr["player_proposing_res"]]
        recv = [int(r) for r in tr["target_player_res"]]
        put("trade_proposer", [int(tr["player_proposing"])])
        put("trade_target", [int(tr["target_player"])])
        put("trade_n_give", [len(give)])
        put("trade_give", give + [0] * (4 - len(give)))
        put("trade_n_recv", [len(recv)])
        put("trade_recv", recv + [0] * (4 - len(recv)))
    put("road_building_active", [int(g.road_building_active[0])])
    put("road_building_count", [int(g.road_building_active[1])])
    put("can_move_robber", [int(g.can_move_robber)])
    put("just_moved_robber", [int(g.just_moved_robber)])
    put("need_discard", [int(g.players_need_to_discard)])
    td = [int(p) for p in g.players_to_discard]
    put("n_to_discard", [len(td)])
    put("to_discard", td + [0] * (4 - len(td)))
    put("die1", [0 if g.die_1 is None else int(g.die_1)])
    put("die2", [0 if g.die_2 is None else int(g.die_2)])
    put("trades_this_turn", [int(g.trades_proposed_this_turn)])
    put("actions_this_turn", [int(g.actions_this_turn)])
    put("turn", [int(g.turn)])
    put("bought_this_turn", [g.development_cards_bought_this_turn.count(DevelopmentCard(c)) for c in range(5)])
    put("lr_player", [0 if g.longest_road is None else int(g.longest_road["player"])])
    put("lr_count", [0 if g.longest_road is None else int(g.longest_road["count"])])
    put("la_player", [0 if g.largest_army is None else int(g.largest_army["player"])])
    put("la_count", [0 if g.largest_army is None else int(g.largest_army["count"])])
    put("cur_longest_path", [int(g.current_longest_path[p]) if p in g.current_longest_path else 0 for p in PIDS])
    put("cur_army_size", [int(g.current_army_size[p]) if p in g.current_army_size else 0 for p in PIDS])
    put("curr_vps", [int(env.curr_vps[p]) for p in PIDS])
    put("winner", [0 if env.winner is None else int(env.winner.id)])
    put("rng_draws", [rng_draws])
    return out


def random_legal_action(masks, env, rng):
    """Uniform-random legal composite action (python-side sampler for fuzzing; richer than the
    device sampler: trades of up to 4+4 resources)."""
    m = [np.asarray(x) for x in masks]
    a = np.zeros((18,), dtype=np.int32)
    types = np.flatnonzero(m[0] > 0)
    t = int(rng.choice(types))
    a[0] = t

    def pick(v):
        idx = np.flatnonzero(np.asarray(v) > 0)
        return int(rng.choice(idx))

    if t == 0:
        a[1] = pick(m[1][0])
    elif t == 2:
        a[1] = pick(m[1][1])
    elif t == 1:
        a[2] = pick(m[2])
    elif t == 8:
        a[3] = pick(m[3])
    elif t == 4:
        a[4] = pick(m[4])
        if a[4] == 4:
            a[15] = pick(m[9][2])
        elif a[4] == 2:
            a[15] = pick(m[9][3])
            a[16] = pick(m[10])
    elif t == 5:
        a[15] = pick(m[9][0])
        a[16] = pick(m[10])
    elif t == 6:
        a[6] = int(rng.integers(0, 3))
        g = env.game
        hand = []
        pl = g.players[g.players_go]
        for i, r in enumerate(RES):
            hand += [i + 1] * int(pl.resources[r])
        n_give = int(rng.integers(1, min(4, len(hand)) + 1))
        give = list(rng.choice(hand, size=n_give, replace=False))
        n_recv = int(rng.integers(1, 5))
        recv = list(rng.integers(1, 6, size=n_recv))
        a[7:7 + n_give] = give
        a[11:11 + n_recv] = recv
    elif t == 7:
        a[5] = pick(m[5])
    elif t == 11:
        a[6] = pick(m[6][1])
    elif t == 12:
        a[17] = pick(m[11])
    return a


# ---------------------------------------------------------------------------------------------------------------------
# Scripted stand-ins for the four nets of a reference worker, so that fixtures produced by the reference's own
# GamesAndPoliciesManager / EvaluationManager do not depend on floating-point details of a network: the decision is a
# seeded random legal action (build-biased, so that games end sooner), recorded in `trace`.
TYPE_WEIGHTS = np.array([60, 8, 60, 12, 12, 3, 1, 1, 6, 1, 1, 1, 1], dtype=float)


def weighted_legal_action(masks, env, rng):
    m0 = np.asarray(masks[0])
    w = TYPE_WEIGHTS * (m0 > 0)
    t = int(rng.choice(13, p=w / w.sum()))
    mm = [np.array(x, copy=True) for x in masks]
    mm[0] = np.zeros(13)
    mm[0][t] = 1
    return random_legal_action(mm, env, rng)


def scripted_log_prob(a18):
    """any deterministic function of the action serves as the stored `action_log_probs`"""
    return -0.25 - float(int(np.sum(a18)) % 7)


class ScriptedContext(object):
    """Which env is deciding right now (set by the hooked `_get_players_turn`), one rng + one Philox stream per env, and the
    trace of every decision (all seats) per env."""

    def __init__(self, envs, streams, rng_seeds):
        self.envs, self.streams = list(envs), list(streams)
        self.rngs = [np.random.default_rng(s) for s in rng_seeds]
        self.trace = [[] for _ in envs]
        self.cur = 0
        for i, env in enumerate(self.envs):          # every draw of env i goes to stream i
            env.reset = self._wrap(env.reset, i)
            env.step = self._wrap(env.step, i)

    def _wrap(self, fn, i):
        def call(*a, **kw):
            with patched_rng(self.streams[i]):
                return fn(*a, **kw)
        return call

    def hook_manager(self, mgr):
        orig = mgr._get_players_turn

        def hooked(env):
            self.cur = self.envs.index(env)
            return orig(env)
        mgr._get_players_turn = hooked


class ScriptedRefPolicy(object):
    """Has the part of the reference net's surface the managers use (`act`, `eval`); returns actions in the net's output
    format (12 entries, [1,1] int64 tensors; heads 7 and 8 lists of four)."""

    def __init__(self, ctx, lstm_size=256):
        self.ctx, self.lstm_size = ctx, lstm_size

    def eval(self):
        return self

    def act(self, obs, hidden_states, terminal_mask, action_masks, deterministic=False):
        import torch
        c = self.ctx
        env = c.envs[c.cur]
        a = weighted_legal_action(env.get_action_masks(), env, c.rngs[c.cur])
        c.trace[c.cur].append(np.array(a, dtype=np.int8))
        heads = action_to_heads(a)
        actions = [[torch.tensor([[int(v)]]) for v in h] if isinstance(h, (list, np.ndarray)) else torch.tensor([[int(h)]]) for h in heads]
        return None, actions, torch.tensor([[scripted_log_prob(a)]], dtype=torch.float32), hidden_states
