"""ctypes binding of the CPU oracle (oracle/libcatan_oracle.so).  Test infrastructure only:
imported by tests/, tools/ (development container), __graft_entry__.smoke() and bench.py's cpu_baseline."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libcatan_oracle.so")
if os.environ.get("CATAN_ORACLE_ASAN"):      # tools/fuzz_oracle_asan.sh: the -fsanitize=address,undefined build (`make -C oracle asan`)
    LIB_PATH = os.path.join(ORACLE_DIR, "libcatan_oracle_asan.so")

STATE_WORDS, MASK_WORDS, OBS_FLOATS, OBS_LISTS, OBS_LIST_PAD, ACTION_WORDS = 736, 325, 1787, 5, 25, 18

_lib = None


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("catan_oracle.c", "catan_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["asan"] if LIB_PATH.endswith("_asan.so") else []),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i32p, f32p, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int64)
        L.orc_env_size.restype = C.c_int
        L.orc_topology.restype = vp
        L.orc_seed_philox.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.orc_seed_mt.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.orc_rng_draws.argtypes = [vp]; L.orc_rng_draws.restype = C.c_uint32
        L.orc_config_default.argtypes = [vp]
        L.orc_set_config.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_double]
        L.orc_set_max_actions_per_turn.argtypes = [vp, C.c_int]
        L.orc_last_reward64.argtypes = [vp, C.POINTER(C.c_double)]
        L.orc_board_reset.argtypes = [vp]
        L.orc_game_reset.argtypes = [vp]
        L.orc_masks.argtypes = [vp, f32p]
        L.orc_action_is_legal.argtypes = [vp, i32p]; L.orc_action_is_legal.restype = C.c_int
        L.orc_action_in_masks.argtypes = [vp, i32p]; L.orc_action_in_masks.restype = C.c_int
        L.orc_step.argtypes = [vp, i32p, f32p, C.POINTER(C.c_int)]; L.orc_step.restype = C.c_int
        L.orc_deciding_player.argtypes = [vp]; L.orc_deciding_player.restype = C.c_int
        L.orc_obs.argtypes = [vp, f32p, i32p, i32p, i32p]
        L.orc_export.argtypes = [vp, i32p]
        L.orc_import.argtypes = [vp, i32p]
        L.orc_longest_path_raw.argtypes = [i32p, i32p, C.c_int]; L.orc_longest_path_raw.restype = C.c_int
        L.orc_sample_action.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, f32p, i32p]
        L.orc_batch_create.argtypes = [vp, C.c_int64, C.c_uint64, C.c_uint64]
        L.orc_batch_run_random.argtypes = [vp, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64, i32p, i64p, C.c_int]
        L.orc_batch_run_random.restype = C.c_int64
        u32p = C.POINTER(C.c_uint32)
        L.orc_batch_run_random_counts.argtypes = [vp, C.c_int64, C.c_uint64, C.c_uint64, u32p, u32p, i32p, i64p, C.c_int]
        L.orc_batch_run_random_counts.restype = C.c_int64
        L.orc_batch_play.argtypes = [vp, C.c_int64, C.c_uint64, C.c_uint64, u32p, C.POINTER(C.c_uint8), i32p, f32p, C.POINTER(C.c_double),
                                     C.POINTER(C.c_uint8), C.c_int]
        L.orc_batch_play.restype = C.c_int64
        L.orc_randomise_uncertainty.argtypes = [vp, C.c_int]
        L.orc_randomise_uncertainty.restype = C.c_int
        L.orc_gae.argtypes = [f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_double, C.c_double, f32p, f32p]
        L.orc_ppo_loss.argtypes = [f32p] * 6 + [C.c_int64, C.c_float, f32p, f32p, f32p, f32p, C.c_float]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class TopologyStruct(C.Structure):
    _fields_ = [("tile_corner", C.c_int * 6 * 19), ("tile_edge", C.c_int * 6 * 19), ("tile_nbr", C.c_int * 6 * 19),
                ("edge_corner", C.c_int * 2 * 72), ("corner_nbr_corner", C.c_int * 3 * 54),
                ("corner_nbr_edge", C.c_int * 3 * 54), ("corner_tile", C.c_int * 3 * 54),
                ("harbour_slot_corner", C.c_int * 2 * 9), ("harbour_slot_edge", C.c_int * 9),
                ("corner_harbour_slot", C.c_int * 54)]


def topology():
    t = TopologyStruct.from_address(lib().orc_topology())
    return {name: np.ctypeslib.as_array(getattr(t, name)).copy() for name, _ in TopologyStruct._fields_}


class OracleEnv(object):
    """One oracle env (philox mode by default)."""

    def __init__(self, seed=0, env_id=0, mt_seeds=None):
        L = lib()
        self.L = L
        self.buf = C.create_string_buffer(L.orc_env_size())
        self.p = C.cast(self.buf, C.c_void_p)
        L.orc_config_default(self.p)
        if mt_seeds is not None:
            L.orc_seed_mt(self.p, mt_seeds[0], mt_seeds[1])
        else:
            L.orc_seed_philox(self.p, seed, env_id)

    def set_config(self, max_trades_per_turn=4, win_reward=500.0, dense_reward=False, reward_annealing_factor=1.0,
                   max_actions_per_turn=None):
        """EnvWrapper keyword arguments (env/wrapper.py:12-13); max_trades_per_turn / max_actions_per_turn None = unlimited"""
        mt = -1 if max_trades_per_turn is None else int(max_trades_per_turn)
        self.L.orc_set_config(self.p, mt, float(win_reward), int(dense_reward), float(reward_annealing_factor))
        self.L.orc_set_max_actions_per_turn(self.p, -1 if max_actions_per_turn is None else int(max_actions_per_turn))

    def last_reward64(self):
        r = np.zeros((4,), dtype=np.float64)
        self.L.orc_last_reward64(self.p, _p(r, C.c_double))
        return r

    def board_reset(self):
        self.L.orc_board_reset(self.p)

    def reset(self):
        self.L.orc_game_reset(self.p)

    def masks(self):
        m = np.zeros((MASK_WORDS,), dtype=np.float32)
        self.L.orc_masks(self.p, _p(m, C.c_float))
        return m

    def is_legal(self, action):
        a = np.ascontiguousarray(action, dtype=np.int32)
        return bool(self.L.orc_action_is_legal(self.p, _p(a, C.c_int32)))

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.int32)
        rew = np.zeros((4,), dtype=np.float32)
        done = C.c_int(0)
        rc = self.L.orc_step(self.p, _p(a, C.c_int32), _p(rew, C.c_float), C.byref(done))
        assert rc == 0
        return rew, bool(done.value)

    def randomise_uncertainty(self, controlling_player):
        """Game.randomise_uncertainty (game.py:1207-1282); returns the attempts of its rejection loop"""
        r = self.L.orc_randomise_uncertainty(self.p, int(controlling_player))
        assert r >= 1
        return r

    def deciding_player(self):
        return self.L.orc_deciding_player(self.p)

    def obs(self):
        f = np.zeros((OBS_FLOATS,), dtype=np.float32)
        lists = np.zeros((OBS_LISTS, OBS_LIST_PAD), dtype=np.int32)
        lens = np.zeros((OBS_LISTS,), dtype=np.int32)
        pid = np.zeros((1,), dtype=np.int32)
        self.L.orc_obs(self.p, _p(f, C.c_float), _p(lists, C.c_int32), _p(lens, C.c_int32), _p(pid, C.c_int32))
        return f, lists, lens, int(pid[0])

    def export(self):
        b = np.zeros((STATE_WORDS,), dtype=np.int32)
        self.L.orc_export(self.p, _p(b, C.c_int32))
        return b

    def import_(self, blob):
        b = np.ascontiguousarray(blob, dtype=np.int32)
        self.L.orc_import(self.p, _p(b, C.c_int32))

    def sample_action(self, seed, env_id, step_idx, masks=None):
        m = self.masks() if masks is None else np.ascontiguousarray(masks, dtype=np.float32)
        a = np.zeros((ACTION_WORDS,), dtype=np.int32)
        self.L.orc_sample_action(self.p, seed, env_id, step_idx, _p(m, C.c_float), _p(a, C.c_int32))
        return a


class OracleBatch(object):
    def __init__(self, n, seed=0, env_id0=0):
        L = lib()
        self.L, self.n, self.seed, self.env_id0 = L, n, seed, env_id0
        self.size = L.orc_env_size()
        self.buf = C.create_string_buffer(self.size * n)
        self.p = C.cast(self.buf, C.c_void_p)
        L.orc_batch_create(self.p, n, seed, env_id0)
        self.step_idx = 0
        self.games = C.c_int64(0)

    def env_ptr(self, i):
        return C.c_void_p(self.p.value + i * self.size)

    def run_random(self, steps, want_blobs=True, n_threads=0):
        blobs = np.zeros((self.n, STATE_WORDS), dtype=np.int32) if want_blobs else None
        self.L.orc_batch_run_random(self.p, self.n, self.seed, self.env_id0, self.step_idx, steps,
                                    _p(blobs, C.c_int32) if want_blobs else None, C.byref(self.games), n_threads)
        self.step_idx += steps
        return blobs

    def run_random_counts(self, counts, start=None, n_threads=0):
        """game i takes counts[i] decisions, its policy stream indexed by its own decision number (start[i] + s)"""
        blobs = np.zeros((self.n, STATE_WORDS), dtype=np.int32)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        start = None if start is None else np.ascontiguousarray(start, dtype=np.uint32)
        self.L.orc_batch_run_random_counts(self.p, self.n, self.seed, self.env_id0,
                                           _p(start, C.c_uint32) if start is not None else None, _p(counts, C.c_uint32),
                                           _p(blobs, C.c_int32), C.byref(self.games), n_threads)
        return blobs

    def play(self, counts, play, actions, reward, reward64, done, n_threads=0):
        """orc_batch_play: one decision (number counts[i], advanced) of the games with play[i]; fills actions / reward / reward64 / done rows"""
        return self.L.orc_batch_play(self.p, self.n, self.seed, self.env_id0, _p(counts, C.c_uint32), _p(play, C.c_uint8), _p(actions, C.c_int32),
                                     _p(reward, C.c_float), _p(reward64, C.c_double), _p(done, C.c_uint8), n_threads)

    def export(self):
        blobs = np.zeros((self.n, STATE_WORDS), dtype=np.int32)
        for i in range(self.n):
            self.L.orc_export(self.env_ptr(i), _p(blobs[i], C.c_int32))
        return blobs

    def import_all(self, blobs):
        blobs = np.ascontiguousarray(blobs, dtype=np.int32)
        assert blobs.shape == (self.n, STATE_WORDS)
        for i in range(self.n):
            self.L.orc_import(self.env_ptr(i), _p(blobs[i], C.c_int32))

    def set_config(self, max_trades_per_turn=4, win_reward=500.0, dense_reward=False, reward_annealing_factor=1.0,
                   max_actions_per_turn=None):
        mt = -1 if max_trades_per_turn is None else int(max_trades_per_turn)
        ma = -1 if max_actions_per_turn is None else int(max_actions_per_turn)
        for i in range(self.n):
            self.L.orc_set_config(self.env_ptr(i), mt, float(win_reward), int(dense_reward), float(reward_annealing_factor))
            self.L.orc_set_max_actions_per_turn(self.env_ptr(i), ma)

    def masks(self):
        m = np.zeros((self.n, MASK_WORDS), dtype=np.float32)
        for i in range(self.n):
            self.L.orc_masks(self.env_ptr(i), _p(m[i], C.c_float))
        return m


def longest_path_raw(edge_owner, corner_owner, player):
    eo = np.ascontiguousarray(edge_owner, dtype=np.int32)
    co = np.ascontiguousarray(corner_owner, dtype=np.int32)
    return lib().orc_longest_path_raw(_p(eo, C.c_int32), _p(co, C.c_int32), int(player))


def gae(rewards, values, masks, gamma, lam):
    T, N = rewards.shape
    r = np.ascontiguousarray(rewards, dtype=np.float32); v = np.ascontiguousarray(values, dtype=np.float32)
    m = np.ascontiguousarray(masks, dtype=np.float32)
    ret = np.zeros((T, N), dtype=np.float32); adv = np.zeros((T, N), dtype=np.float32)
    lib().orc_gae(_p(r, C.c_float), _p(v, C.c_float), _p(m, C.c_float), T, N, gamma, lam, _p(ret, C.c_float), _p(adv, C.c_float))
    return ret, adv


def ppo_loss(logp, old_logp, adv, values, old_values, returns, clip, value_coef=1.0):
    arrs = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1) for x in (logp, old_logp, adv, values, old_values, returns)]
    B = arrs[0].shape[0]
    la = C.c_float(0); lv = C.c_float(0)
    dl = np.zeros((B,), dtype=np.float32); dv = np.zeros((B,), dtype=np.float32)
    lib().orc_ppo_loss(*[_p(x, C.c_float) for x in arrs], B, clip, C.byref(la), C.byref(lv), _p(dl, C.c_float), _p(dv, C.c_float), value_coef)
    return la.value, lv.value, dl, dv
