"""Host-side mirror of the reference env interface over libcatan_hip.so.

* `VecCatanEnv`  - N games on one GPU; torch tensors in, torch tensors out (zero-copy device pointers).
* `EnvWrapper`   - single-game view with the reference's signatures (`reset() -> obs`, `step(action) ->
                   (obs, reward_dict, done, info)`, `get_action_masks() -> list of 12 np.ndarray`,
                   `save_state()/restore_state()`), reference env/wrapper.py:11-50,168-185,711-721, so that
                   RL/ppo/game_manager.py style callers keep working.
PyTorch is only plumbing here (device memory + streams); all game logic runs in the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, spec


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class VecCatanEnv(object):
    def __init__(self, num_envs, seed=0, env_id0=0, device=None, max_proposed_trades_per_turn=4, win_reward=500.0,
                 dense_reward=False, validate_actions=True, auto_reset=True, max_actions_per_turn=None):
        if not torch.cuda.is_available():
            raise _lib.CatanHipError("VecCatanEnv needs a HIP device (no CPU fallback)")
        self.L = _lib.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = int(num_envs)
        cfg = _lib.CatanCfg()
        self.L.catan_cfg_default(C.byref(cfg))
        cfg.max_proposed_trades_per_turn = -1 if max_proposed_trades_per_turn is None else int(max_proposed_trades_per_turn)
        cfg.win_reward = float(win_reward)
        cfg.dense_reward = int(bool(dense_reward))
        cfg.validate_actions = int(bool(validate_actions))
        cfg.auto_reset = int(bool(auto_reset))
        if max_actions_per_turn is not None and (max_actions_per_turn != max_actions_per_turn or max_actions_per_turn < 0):
            raise ValueError("max_actions_per_turn must be None (no limit) or a non-negative number")
        # env/wrapper.py:14-17,233: None -> np.inf; the test is `actions_this_turn > max_actions_per_turn` on an integer counter
        cfg.max_actions_per_turn = -1 if (max_actions_per_turn is None or max_actions_per_turn == float("inf")) \
            else int(min(max_actions_per_turn // 1, 2 ** 31 - 1))
        self.cfg = cfg
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.catan_create(C.byref(h), self.device.index or 0, self.n, seed, env_id0, C.byref(cfg)))
        self.h = h
        self.reward = torch.zeros((self.n, 4), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((self.n,), dtype=torch.uint8, device=self.device)
        self.reward64 = None

    def set_step_wave_games(self, games):
        """scheduling knob of k_step: 64 / 32 / 16 games per wave (results do not depend on it)"""
        _lib.check(self.L.catan_set_step_wave_games(self.h, int(games)))

    def enable_reward64(self):
        """Every later step also leaves its rewards UNROUNDED (the reference's Python floats, env/wrapper.py:85-112) in
        `self.reward64` (float64 [n][4]); rollout collection sums those over a turn before rounding, as
        RL/ppo/game_manager.py:94-95 does."""
        if self.reward64 is None:
            self.reward64 = torch.zeros((self.n, 4), dtype=torch.float64, device=self.device)
            _lib.check(self.L.catan_set_reward_f64_buffer(self.h, _ptr(self.reward64)))
        return self.reward64

    def close(self):
        if getattr(self, "h", None):
            self.L.catan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference-shaped calls, batched
    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self.L.catan_reset(self.h, _ptr(m), _stream()))

    # ---- RNG contract (A): a single-game handle on the reference's two Mersenne Twisters (include/catan_hip.h)
    def seed_mt19937(self, numpy_seed, python_seed):
        """as `np.random.seed(numpy_seed); random.seed(python_seed)` in front of the reference's EnvWrapper"""
        _lib.check(self.L.catan_seed_mt19937(self.h, int(numpy_seed) & 0xFFFFFFFF, int(python_seed) & 0xFFFFFFFF, _stream()))

    def mt19937_from_process_globals(self):
        """installs COPIES of this process's np.random / random generator states (np.random.get_state(), random.getstate()): what
        `np.random.seed(s); random.seed(s)` left there.  The library's kernels advance the copies; the process's own generators stay put."""
        import random
        kind, key, pos = np.random.get_state()[:3]
        if kind != "MT19937":
            raise ValueError("np.random is not the legacy MT19937 RandomState")
        key = np.ascontiguousarray(key, dtype=np.uint32)
        _lib.check(self.L.catan_mt19937_set_state(self.h, 0, key.ctypes.data_as(C.c_void_p), int(pos), _stream()))
        st = random.getstate()[1]
        pkey = np.array(st[:624], dtype=np.uint32)
        _lib.check(self.L.catan_mt19937_set_state(self.h, 1, pkey.ctypes.data_as(C.c_void_p), int(st[624]), _stream()))

    def reset_board_only(self):
        """Board.reset alone: the draws of the reference's Board() constructor (game/components/board.py:47)"""
        _lib.check(self.L.catan_reset_board_only(self.h, _stream()))

    def step(self, actions):
        """actions: int32 [n][18] (a negative type = no-op).  Returns (reward [n][4] float32 indexed by PlayerId-1,
        done [n] uint8) - views of buffers owned by the env, overwritten by the next step."""
        a = actions.to(device=self.device, dtype=torch.int32).contiguous()
        assert a.shape == (self.n, spec.ACTION_WORDS), a.shape
        _lib.check(self.L.catan_step(self.h, _ptr(a), _ptr(self.reward), _ptr(self.done), _stream()))
        return self.reward, self.done

    STEP_COMPLETE, STEP_WAITING, STEP_NONE = 0, 1, 2       # include/catan_hip.h

    def step_deferred(self, actions, window=32, status_out=None):
        """catan_step_deferred (include/catan_hip.h): the same actions as `step`, but a game whose step needs the slow path
        (longest road, re-deal) completes it on side streams and WAITS meanwhile.  Returns (reward, done, status): status[g] == 0:
        reward / done are the result of game g's last applied action and its state is current; 1: game g is waiting (its entry
        of the next call's actions is ignored, its state must not be read).  `step_flush()` closes the sequence."""
        a = actions if (actions.dtype == torch.int32 and actions.is_contiguous() and actions.device == self.device) else \
            actions.to(device=self.device, dtype=torch.int32).contiguous()
        assert a.shape == (self.n, spec.ACTION_WORDS), a.shape
        if status_out is None:
            if getattr(self, "status", None) is None:
                self.status = torch.zeros((self.n,), dtype=torch.uint8, device=self.device)
            status_out = self.status
        _lib.check(self.L.catan_step_deferred(self.h, _ptr(a), int(window), _ptr(self.reward), _ptr(self.done), _ptr(status_out), _stream()))
        return self.reward, self.done, status_out

    def step_flush(self):
        """catan_step_flush: completes every outstanding deferred step.  status 0: a game that was waiting (reward / done valid);
        2: nothing was outstanding for it."""
        if getattr(self, "status", None) is None:
            self.status = torch.zeros((self.n,), dtype=torch.uint8, device=self.device)
        _lib.check(self.L.catan_step_flush(self.h, _ptr(self.reward), _ptr(self.done), _ptr(self.status), _stream()))
        return self.reward, self.done, self.status

    def get_action_masks(self, out=None, games=None):
        """float32 [n][325]; slice with spec.MASK_OFFSETS / MASK_SHAPES for the 12 heads.  games (int32 [k]): catan_masks_of - row j
        = the masks of game games[j] (the first k rows of `out`)."""
        if out is None:
            out = torch.empty((self.n if games is None else games.numel(), spec.MASK_WORDS), dtype=torch.float32, device=self.device)
        if games is not None:
            _lib.check(self.L.catan_masks_of(self.h, _ptr(out), _ptr(games), games.numel(), _stream()))
            return out
        _lib.check(self.L.catan_masks(self.h, _ptr(out), _stream()))
        return out

    def get_action_masks_packed(self):
        """int32 [n][11]: the same masks as 325-bit rows (bit i of the flat mask = word i >> 5, bit i & 31)"""
        out = torch.empty((self.n, 11), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_masks_packed_copy(self.h, _ptr(out), _stream()))
        return out

    def get_action_masks_by_head(self):
        flat = self.get_action_masks()
        return [flat[:, o:o + s].reshape((self.n,) + shp)
                for o, s, shp in zip(spec.MASK_OFFSETS, spec.MASK_SIZES, spec.MASK_SHAPES)]

    def deciding_player(self):
        out = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_deciding_seat(self.h, _ptr(out), _stream()))
        return out

    def players_turn_sim(self):
        """worker.py:146-151: trade target > players_go, ignoring the discard phase (forward-search simulations)"""
        out = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_players_turn_sim(self.h, _ptr(out), _stream()))
        return out

    def sample_random_actions(self, step_idx, out=None):
        if out is None:
            out = torch.empty((self.n, spec.ACTION_WORDS), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_sample_random_actions(self.h, int(step_idx), _ptr(out), _stream()))
        return out

    def random_rollout(self, step_idx0, steps):
        _lib.check(self.L.catan_random_rollout(self.h, int(step_idx0), int(steps), _stream()))

    def randomise_uncertainty(self, controlling_player):
        """Game.randomise_uncertainty for every game with controlling_player[i] in 1..4 (0: untouched), game.py:1207-1282"""
        cp = torch.as_tensor(controlling_player, device=self.device).to(torch.int32).contiguous()
        assert cp.shape == (self.n,)
        _lib.check(self.L.catan_randomise_uncertainty(self.h, _ptr(cp), _stream()))

    def missed_speculation_count(self):
        """finished games of lock-step steps that had no speculatively dealt successor (expected 0)"""
        return int(self.L.catan_missed_speculation_count(self.h, _stream()))

    def inconsistent_deal_count(self):
        return int(self.L.catan_inconsistent_deal_count(self.h, _stream()))

    def random_rollout_deferred(self, iters, window):
        """`iters` iterations of the deferred loop (include/catan_hip.h): games that need the slow path (longest road,
        re-deal) sit out until their window of `window` iterations closes; per-game trajectories are the lock-step ones."""
        _lib.check(self.L.catan_random_rollout_deferred(self.h, int(iters), int(window), _stream()))

    def policy_counters(self):
        """-> int64 [n]: decisions taken by every game in deferred rollouts (its policy-stream index)"""
        out = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_policy_counters(self.h, _ptr(out), _stream()))
        return out.long() & 0xFFFFFFFF

    def set_policy_counters(self, counters=None):
        c = None if counters is None else torch.as_tensor(counters, device=self.device).to(torch.int32).contiguous()
        _lib.check(self.L.catan_set_policy_counters(self.h, _ptr(c), _stream()))

    def set_deferred_fused(self, on):
        """which form of the deferred loop runs (include/catan_hip_tuning.h): results are identical, game for game"""
        _lib.check(self.L.catan_set_deferred_fused(self.h, int(bool(on))))

    @property
    def deferred_fused(self):
        return bool(self.L.catan_deferred_fused(self.h))

    def set_lr_budgets(self, lockstep, deferred):
        _lib.check(self.L.catan_set_lr_budgets(self.h, int(lockstep), int(deferred)))

    def slow_path_counts(self):
        """-> cumulative (tier-1 longest-road requests, requests handed to tier 2, k_lr_finish launches)"""
        out = (C.c_uint64 * 3)()
        _lib.check(self.L.catan_slow_path_counts(self.h, _stream(), out))
        return int(out[0]), int(out[1]), int(out[2])

    def set_lr_rounds(self, lockstep, deferred):
        _lib.check(self.L.catan_set_lr_rounds(self.h, int(lockstep), int(deferred)))

    def random_rollout_timed(self, step_idx0, steps, window=0):
        """-> dict of summed per-kernel milliseconds (hipEvents on the launch stream); window > 0: the deferred loop."""
        ms = (C.c_float * 5)()
        _lib.check(self.L.catan_random_rollout_timed(self.h, int(step_idx0), int(steps), int(window), _stream(), ms))
        return dict(zip(("k_sample_random", "k_step", "k_lr_finish", "k_lr_heavy", "k_reset_list"), [float(x) for x in ms]))

    def export_state(self, env_idx=None):
        """-> int32 [cnt][736] canonical blobs (host-friendly orientation)."""
        idx = None
        cnt = self.n
        if env_idx is not None:
            idx = torch.as_tensor(env_idx, dtype=torch.int64, device=self.device).contiguous()
            cnt = idx.numel()
        blob = torch.empty((spec.STATE_WORDS, cnt), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_state_export(self.h, _ptr(blob), _ptr(idx), cnt, _stream()))
        return blob.t().contiguous()

    def import_state(self, blobs, env_idx=None):
        b = blobs if torch.is_tensor(blobs) else torch.as_tensor(np.asarray(blobs))
        b = b.to(device=self.device, dtype=torch.int32)
        if b.dim() == 1:
            b = b[None]
        cnt = b.shape[0]
        idx = None
        if env_idx is not None:
            idx = torch.as_tensor(env_idx, dtype=torch.int64, device=self.device).contiguous()
            assert idx.numel() == cnt
        bt = b.t().contiguous()
        _lib.check(self.L.catan_state_import(self.h, _ptr(bt), _ptr(idx), cnt, _stream()))

    def get_obs(self, out=None):
        from . import obs as _obs
        return _obs.get_obs(self, out)

    def get_obs_rows(self, dtype=torch.float32, out=None, rows=None, t=None, sel=None, dense=True, games=None):
        from . import obs as _obs
        return _obs.get_obs_rows(self, dtype, out, rows, t, sel, dense, games)

    def longest_path(self, players):
        """Game.get_longest_path for PlayerId players[i] of game i (diagnostic/test entry)."""
        p = torch.as_tensor(players, dtype=torch.int32, device=self.device).contiguous()
        out = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        _lib.check(self.L.catan_longest_path(self.h, _ptr(p), _ptr(out), _stream()))
        return out

    def invalid_action_count(self):
        return int(self.L.catan_invalid_action_count(self.h, _stream()))

    def set_reward_annealing_factor(self, f):
        _lib.check(self.L.catan_set_reward_annealing(self.h, float(f)))


class _GameView(object):
    """The `env.game.*` attributes the reference's callers read: players_need_to_discard,
    players_to_discard, must_respond_to_trade, proposed_trade["target_player"], players_go (game_manager.py:152-159),
    initial_settlements_placed / initial_roads_placed (evaluation/evaluation_manager.py:84-88) and
    randomise_uncertainty(player_id) (forward_search_policy/worker.py:46)."""

    def __init__(self, wrapper):
        self._w = wrapper

    def _field(self, name):
        return spec.state_field(self._w._blob(), name)

    @property
    def players_need_to_discard(self):
        return bool(self._field("need_discard")[0])

    @property
    def players_to_discard(self):
        n = int(self._field("n_to_discard")[0])
        return [int(x) for x in self._field("to_discard")[:n]]

    @property
    def must_respond_to_trade(self):
        return bool(self._field("must_respond")[0])

    @property
    def proposed_trade(self):
        if not self.must_respond_to_trade:
            return None
        ng, nr = int(self._field("trade_n_give")[0]), int(self._field("trade_n_recv")[0])
        return {"player_proposing": int(self._field("trade_proposer")[0]),
                "target_player": int(self._field("trade_target")[0]),
                "player_proposing_res": [int(x) for x in self._field("trade_give")[:ng]],
                "target_player_res": [int(x) for x in self._field("trade_recv")[:nr]]}

    @property
    def players_go(self):
        return int(self._field("players_go")[0])

    @property
    def initial_placement_phase(self):
        return bool(self._field("initial_phase")[0])

    @property
    def initial_settlements_placed(self):
        v = self._field("init_settlements")
        return {pid: int(v[pid - 1]) for pid in (1, 2, 3, 4)}

    @property
    def initial_roads_placed(self):
        v = self._field("init_roads")
        return {pid: int(v[pid - 1]) for pid in (1, 2, 3, 4)}

    def randomise_uncertainty(self, controlling_player_id):
        self._w.vec.randomise_uncertainty(torch.tensor([int(controlling_player_id)]))
        self._w._cache = None


class EnvWrapper(object):
    """Single-game view with the reference EnvWrapper signatures (env/wrapper.py:11-50)."""

    def __init__(self, interactive=False, max_actions_per_turn=None, max_proposed_trades_per_turn=4, validate_actions=True,
                 debug_mode=False, win_reward=500, dense_reward=False, policies=None, seed=0, env_id=0, rng="philox"):
        """The reference's keyword arguments in the reference's order (env/wrapper.py:12-13) plus the game's Philox stream
        (seed, env_id).  Anything else is a TypeError, as with the reference; `interactive` / `debug_mode` / `policies` drive
        the reference's pygame display and its text log (game/game.py:30-37), which are out of scope: only their defaults.
        rng="mt19937": RNG contract (A) - the game draws from copies of THIS PROCESS's np.random / random generators as they are now
        (`np.random.seed(s); random.seed(s); env = EnvWrapper(rng="mt19937")` replays the unpatched reference draw for draw,
        tests/test_gpu_golden.py::test_mt19937_known_answer_on_the_hip_path); the constructor takes the draws of Board() and Game()."""
        if interactive or debug_mode or policies is not None:
            raise NotImplementedError("EnvWrapper(interactive / debug_mode / policies): the reference's display and text log "
                                      "are not part of the batched HIP path")
        self.vec = VecCatanEnv(1, seed=seed, env_id0=env_id, max_proposed_trades_per_turn=max_proposed_trades_per_turn,
                               win_reward=win_reward, dense_reward=dense_reward, validate_actions=validate_actions,
                               auto_reset=False, max_actions_per_turn=max_actions_per_turn)
        self.max_actions_per_turn = float("inf") if max_actions_per_turn is None else max_actions_per_turn
        self.max_proposed_trades_per_turn = max_proposed_trades_per_turn
        self.win_reward, self.dense_reward = win_reward, dense_reward
        self.vec.enable_reward64()       # step() hands back the reference's Python floats (unrounded doubles)
        self.validate_actions = validate_actions
        self.game = _GameView(self)
        self._cache = None
        self._fresh = True          # catan_create already reset the game: the first reset() must not draw again
        if rng == "mt19937":
            self.vec.mt19937_from_process_globals()
            self.vec.reset_board_only()          # Board() (game/components/board.py:47)
            self.vec.reset()                     # Game.__init__ -> reset (game/game.py:40)
            self._fresh = False                  # EnvWrapper.reset() deals again, as the reference's does
        elif rng != "philox":
            raise ValueError("rng: 'philox' or 'mt19937'")
        self._reward_annealing_factor = 1.0

    @property
    def reward_annealing_factor(self):
        return self._reward_annealing_factor

    @reward_annealing_factor.setter
    def reward_annealing_factor(self, f):
        self._reward_annealing_factor = f
        self.vec.set_reward_annealing_factor(f)

    def _blob(self):
        if self._cache is None:
            self._cache = self.vec.export_state()[0].cpu().numpy()
        return self._cache

    def reset(self):
        if self._fresh:
            self._fresh = False
        else:
            self.vec.reset()
        self._cache = None
        return self._get_obs()

    def get_action_masks(self):
        flat = self.vec.get_action_masks()[0].cpu().numpy().astype(np.float64)
        return [flat[o:o + s].reshape(shp).copy()
                for o, s, shp in zip(spec.MASK_OFFSETS, spec.MASK_SIZES, spec.MASK_SHAPES)]

    def step(self, action):
        flat = np.zeros((spec.ACTION_WORDS,), dtype=np.int32)
        for h, (off, ln) in enumerate(spec.ACTION_HEAD_SLICES):
            v = np.asarray(action[h]).reshape(-1)
            flat[off:off + min(ln, len(v))] = v[:ln]
        bad0 = self.vec.invalid_action_count() if self.validate_actions else 0
        reward, done = self.vec.step(torch.from_numpy(flat).view(1, spec.ACTION_WORDS))
        self._cache = None
        if self.validate_actions and self.vec.invalid_action_count() != bad0:
            raise RuntimeError("invalid action (Game.validate_action rejects it; game/game.py:264-525)")   # reference env/wrapper.py:38-41
        r = self.vec.reward64[0].cpu().numpy()
        rew = {pid: float(r[pid - 1]) for pid in (1, 2, 3, 4)}
        return self._get_obs(), rew, bool(done[0].item()), {"log": None}

    def _get_obs(self):
        from . import obs as _obs   # the observation encoder is a separate kernel + binding
        return _obs.single_env_obs(self.vec)

    def save_state(self):
        return {"blob": self._blob().copy()}

    def restore_state(self, state):
        self.vec.import_state(state["blob"][None])
        self._cache = None

    @property
    def winner(self):
        w = int(spec.state_field(self._blob(), "winner")[0])
        return None if w == 0 else type("Winner", (), {"id": w})()

    @property
    def curr_vps(self):
        v = spec.state_field(self._blob(), "curr_vps")
        return {pid: int(v[pid - 1]) for pid in (1, 2, 3, 4)}
