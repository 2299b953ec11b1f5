"""CPU stand-in for VecCatanEnv over the oracle (test infrastructure): same duck-typed interface, torch CPU tensors."""
import ctypes as C

import numpy as np
import torch

import oracle_lib


class OracleVecEnv(object):
    def __init__(self, n, seed=0, env_id0=0, dense_reward=False, auto_reset=True):
        self.n, self.seed, self.env_id0 = n, seed, env_id0
        self.device = torch.device("cpu")
        self.b = oracle_lib.OracleBatch(n, seed, env_id0)
        self.L = self.b.L
        self.steps_taken = np.zeros(n, dtype=np.int64)
        self.auto_reset = auto_reset
        self.dense_reward = bool(dense_reward)
        if dense_reward:
            self.b.set_config(dense_reward=True)
        self.reward64 = None
        self.annealing_log = []

    def set_reward_annealing_factor(self, f):
        self.annealing_log.append(float(f))
        self.b.set_config(dense_reward=self.dense_reward, reward_annealing_factor=float(f))

    def enable_reward64(self):
        if self.reward64 is None:
            self.reward64 = torch.zeros((self.n, 4), dtype=torch.float64)
        return self.reward64

    def export_state(self):
        return torch.from_numpy(self.b.export())

    def import_state(self, blobs):
        self.b.import_all(np.asarray(blobs))

    def advance_random(self, steps):
        self.b.run_random(steps, want_blobs=False)

    def deciding_player(self):
        return torch.tensor([self.L.orc_deciding_player(self.b.env_ptr(i)) for i in range(self.n)], dtype=torch.int32)

    def players_turn_sim(self):
        return torch.tensor([self.L.orc_players_turn_sim(self.b.env_ptr(i)) for i in range(self.n)], dtype=torch.int32)

    def randomise_uncertainty(self, ctrl):
        for i, c in enumerate(np.asarray(ctrl).tolist()):
            if c:
                self.L.orc_randomise_uncertainty(self.b.env_ptr(i), int(c))

    def get_obs(self):
        n = self.n
        f = np.zeros((n, 1787), dtype=np.float32); lists = np.zeros((n, 5, 25), dtype=np.int32)
        lens = np.zeros((n, 5), dtype=np.int32); pid = np.zeros((1,), dtype=np.int32)
        for i in range(n):
            self.L.orc_obs(self.b.env_ptr(i), f[i].ctypes.data_as(C.POINTER(C.c_float)), lists[i].ctypes.data_as(C.POINTER(C.c_int32)),
                           lens[i].ctypes.data_as(C.POINTER(C.c_int32)), pid.ctypes.data_as(C.POINTER(C.c_int32)))
        return torch.from_numpy(f), torch.from_numpy(lists), torch.from_numpy(lens)

    def get_action_masks(self):
        return torch.from_numpy(self.b.masks())

    def step(self, actions):
        a = actions.numpy().astype(np.int32)
        rew = np.zeros((self.n, 4), dtype=np.float32); done = np.zeros((self.n,), dtype=np.uint8)
        for i in range(self.n):
            if a[i, 0] < 0:
                if self.reward64 is not None:
                    self.reward64[i] = 0.0
                continue
            ai = np.ascontiguousarray(a[i]); r = np.zeros(4, dtype=np.float32); d = C.c_int(0)
            assert self.L.orc_action_is_legal(self.b.env_ptr(i), ai.ctypes.data_as(C.POINTER(C.c_int32))), (i, ai)
            self.L.orc_step(self.b.env_ptr(i), ai.ctypes.data_as(C.POINTER(C.c_int32)), r.ctypes.data_as(C.POINTER(C.c_float)), C.byref(d))
            rew[i] = r; done[i] = d.value
            if self.reward64 is not None:
                r64 = np.zeros(4, dtype=np.float64)
                self.L.orc_last_reward64(self.b.env_ptr(i), r64.ctypes.data_as(C.POINTER(C.c_double)))
                self.reward64[i] = torch.from_numpy(r64)
            self.steps_taken[i] += 1
            if d.value and self.auto_reset:
                self.L.orc_game_reset(self.b.env_ptr(i))
        return torch.from_numpy(rew), torch.from_numpy(done)


class ScriptedPolicy(object):
    """Uniform-random legal policy keyed by (game, number of decisions that game has taken): independent of the order in
    which games are evaluated, so a lock-step collector and a sequential one make identical decisions."""

    def __init__(self, env):
        self.env = env

    def act(self, f, lists, lens, masks, generator=None, deterministic=False, idx=None):
        env = self.env
        n = f.shape[0]
        assert n == env.n
        a = np.zeros((n, 18), dtype=np.int64)
        for i in range(n):
            m = np.ascontiguousarray(masks[i].numpy(), dtype=np.float32)
            out = np.zeros(18, dtype=np.int32)
            env.L.orc_sample_action(env.b.env_ptr(i), env.seed + 99, env.env_id0 + i, int(env.steps_taken[i]),
                                    m.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_int32)))
            a[i] = out
        logp = torch.from_numpy(-(a.sum(1) % 7).astype(np.float32))[:, None]     # any deterministic stand-in for log-probs
        return torch.zeros(n, 1), torch.from_numpy(a), logp


class RecurrentScriptedPolicy(ScriptedPolicy):
    """ScriptedPolicy plus a toy recurrent state with the LSTM interface of policy.CatanPolicy(include_lstm=True): the
    state is a deterministic function of (incoming state x terminal mask, observation, action), so bookkeeping mistakes
    (wrong seat, missed reset at a game end, state advanced for a frozen game) change it."""
    include_lstm, lstm_size = True, 3

    def act(self, f, lists, lens, masks, generator=None, deterministic=False, idx=None, hidden=None, nonterminal=None):
        v, a, lp = super().act(f, lists, lens, masks)
        h, c = hidden
        nt = nonterminal.reshape(-1, 1).float()
        feat = torch.stack((f[:, :40].sum(1) % 5.0, a[:, 0].float(), torch.ones(f.shape[0])), 1)
        return v, a, lp, (0.5 * h * nt + feat, c * nt + 1.0)
