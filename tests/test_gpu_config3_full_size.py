"""BASELINE.json configs[2] AT ITS OWN SIZE under `-m gpu`: 65 536 games x T = 200 on the DEFAULT collector (catan_step_deferred with
window 4, bucketed hipGraph policy passes over the games that still miss observations), after a 1 500-step random pre-roll, two
consecutive gather_rollouts - so games end INSIDE a rollout and carry over into the next one (RL/ppo/game_manager.py:99-136,
142-150) - with the bf16 RL/models net sampling the decisions.

What is checked:
  * all 65 536 games: every game holds T + 1 observations / T actions, no action was rejected, terminal masks are 0 / 1,
    `games_complete` grew by the number of terminal-mask zeros the rollout stored, log-probs finite and <= 0;
  * 64 sampled games (>= 8 of them must have ENDED inside a rollout): the decisions of ALL FOUR SEATS are logged on the device (no host
    read per iteration: the collector runs its real asynchronous schedule) and replayed on the CPU oracle from the exported pre-roll
    states through a per-game restatement of the reference's list bookkeeping (game_manager.py:69-140); the stored observations, list
    observations, action masks, terminal masks, rewards (summed over the other seats' moves) and actions of both rollouts must be the
    oracle's, and so must the games' final states;
  * the log-prob the rollout stored against the learner's evaluate_actions of the same (observation, action) rows under bf16 autocast
    (the collector's policy pass sums part products that the learner computes as one; tolerance stated below).
Also: four full-size env + collector + storage sets created and destroyed in one process (the round-4 crash of
tools/rollout_schedules.py)."""
import gc

import os

import numpy as np
import pytest
import torch

from settlers_of_catan_rl_amd import spec

pytestmark = pytest.mark.gpu

LOGP_ROLLOUT_VS_LEARNER_TOL = 0.02       # |stored log-prob - evaluate_actions| of a composite action, bf16 autocast on both sides


class DeviceRecorder(object):
    """Delegates to the env and logs, ON THE DEVICE, what the sampled games were given in every env call of the collector: the action
    row and (deferred stepping) whether the game was WAITING when the call was made, i.e. the row was ignored."""

    def __init__(self, env, sample, max_calls):
        self.env, self.n, self.device = env, env.n, env.device
        self.idx = torch.as_tensor(sample, device=env.device, dtype=torch.int64)
        k = self.idx.numel()
        self.log_a = torch.full((max_calls, k, spec.ACTION_WORDS), -7, dtype=torch.int32, device=env.device)
        self.log_w = torch.zeros((max_calls, k), dtype=torch.uint8, device=env.device)
        self.calls, self.marks = 0, []
        self._prev_status = None

    def mark(self):
        """a rollout boundary"""
        self.marks.append(self.calls)

    def step(self, actions):
        self.log_a[self.calls] = actions[self.idx].to(torch.int32)
        self.calls += 1
        return self.env.step(actions)

    def step_deferred(self, actions, window=32, status_out=None):
        self.log_a[self.calls] = actions[self.idx].to(torch.int32)
        if self._prev_status is not None:
            self.log_w[self.calls] = (self._prev_status[self.idx] == 1).to(torch.uint8)
        self.calls += 1
        r, d, s = self.env.step_deferred(actions, window, status_out=status_out)
        self._prev_status = s                  # (the collector alternates two status rows: this one is not rewritten before the next call)
        return r, d, s

    def step_flush(self):
        self._prev_status = None
        return self.env.step_flush()

    def traces(self):
        """-> per rollout, per sampled game: the list of action rows the game really applied"""
        a, w = self.log_a[:self.calls].cpu().numpy(), self.log_w[:self.calls].cpu().numpy()
        out, lo = [], 0
        for hi in self.marks:
            out.append([[a[c, j] for c in range(lo, hi) if a[c, j, 0] >= 0 and not w[c, j]] for j in range(a.shape[1])])
            lo = hi
        return out

    def __getattr__(self, name):
        return getattr(self.env, name)


class GameLists(object):
    """One game of GamesAndPoliciesManager (game_manager.py:35-59,69-150) on an oracle env, fed with a recorded trace."""

    def __init__(self, o, active, T):
        self.o, self.active, self.T = o, active, T
        self.obs, self.masks, self.acts, self.amasks, self.rews = [], [1.0], [], [], []
        self.ends = 0
        if o.deciding_player() == active:                                   # :43-52
            self.obs.append(self._obs())

    def _obs(self):
        f, lists, lens, _ = self.o.obs()
        return f, lists, lens

    def gather(self, trace):
        o, act, T = self.o, self.active, self.T
        racc, done_since, used = np.zeros(4), False, 0
        while len(self.obs) < T + 1:                                        # :78
            assert used < len(trace), "the recorded trace ends before the game holds its T + 1 observations"
            a = trace[used]; used += 1
            dec = o.deciding_player()
            m = o.masks()
            assert o.is_legal(a), (a,)
            o.step(a)
            racc += o.r64                                                   # :94-95 (Python floats)
            done = o.done
            if dec == act:                                                  # :102-105
                self.acts.append(np.asarray(a)); self.amasks.append(m)
            ndec = o.deciding_player()          # (after the reset of a finished game, as the collector reads it: see test_rollout_cpu)
            if done:                                                        # :112-118
                self.rews.append(racc[act - 1]); racc[:] = 0.0
                self.masks.append(0.0); done_since = False; self.ends += 1
            elif ndec == act and len(self.acts) > 0 and not done_since:     # :106-110
                self.rews.append(racc[act - 1]); racc[act - 1] = 0.0
            if ndec == act:                                                 # :126-133
                if not done and not done_since:
                    self.masks.append(1.0)
                done_since = False
                self.obs.append(self._obs())
            elif done:
                done_since = True
        return used

    def after(self):                                                        # :142-150
        self.obs, self.masks = [self.obs[-1]], [self.masks[-1]]
        self.acts, self.amasks, self.rews = [], [], []


class AutoResetGame(object):
    """an oracle game behind `EnvWrapper.step` + the manager's reset of a finished game (game_manager.py:113): keeps the last
    step's done flag and unrounded rewards, which the re-deal would wipe"""

    def __init__(self, oracle, seed, gidx, blob):
        self.o = oracle.OracleEnv(seed, gidx)
        self.o.import_(blob)
        self.done, self.r64 = False, np.zeros(4)

    def step(self, a):
        _, self.done = self.o.step(a)
        self.r64 = self.o.last_reward64().copy()
        if self.done:
            self.o.reset()

    def __getattr__(self, name):
        return getattr(self.o, name)


def test_config3_full_size_default_collector_two_rollouts_sampled_oracle_parity(oracle, hip_lib):
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    n, T, seed, gathers = 65536, 200, 31, 2
    env = VecCatanEnv(n, seed=seed)
    env.random_rollout(0, 1500)                                             # games of every age; many are close to their end
    g = np.random.RandomState(7)
    sample = sorted(set([0, 1, 63, 64, 4095, 32768, 65535] + g.randint(0, n, 57).tolist()))
    blobs0 = env.export_state(sample).cpu().numpy()
    rec = DeviceRecorder(env, sample, max_calls=3200)
    torch.manual_seed(1)
    net = CatanPolicy().cuda()
    col = RolloutCollector(rec, net, T, seed=5, autocast_dtype=torch.bfloat16)
    assert col.deferred_window == 4 and col.graph_act and len(col._bucket_list()) > 4, "this test is about the DEFAULT schedule"
    sidx = torch.as_tensor(sample, device=env.device)
    snaps, complete0 = [], 0
    for r in range(gathers):
        st = col.gather_rollouts()
        rec.mark()
        # ---- all games
        assert env.invalid_action_count() == 0
        assert bool((col.n_obs == T + 1).all()) and bool((col.n_act == T).all()) and bool((col.n_rew >= T).all())
        tm = st.masks[:T + 2]
        assert bool(((tm == 0) | (tm == 1)).all())
        slot = torch.arange(T + 2, device=env.device)[:, None]
        zeros = int(((tm == 0) & (slot >= 1) & (slot < col.n_msk[None, :])).sum())
        assert st.games_complete - complete0 == zeros, (st.games_complete, complete0, zeros)
        complete0 = st.games_complete
        assert bool(torch.isfinite(st.action_log_probs).all()) and bool((st.action_log_probs <= 0).all())
        assert len(col.bucket_log) >= 4, col.bucket_log                     # the games froze into smaller policy passes on the way
        snaps.append({k: getattr(st, k)[:, sidx].cpu() for k in ("obs_f", "lists", "lens", "masks", "rewards", "actions", "action_masks", "action_log_probs")})
        snaps[-1]["am_float"] = st.unpack_action_masks(st.action_masks[:, sidx]).cpu()
        if r == gathers - 1:
            rows_f, rows_l, rows_n = st.obs_f[:T, sidx].reshape(-1, spec.OBS_FLOATS), st.lists[:T, sidx].reshape(-1, 5, spec.OBS_LIST_PAD), st.lens[:T, sidx].reshape(-1, 5)
            with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                ev = net.evaluate_actions(rows_f, rows_l, rows_n.long(), st.unpack_action_masks(st.action_masks[:, sidx]).reshape(-1, spec.MASK_WORDS),
                                          st.actions[:, sidx].reshape(-1, spec.ACTION_WORDS))
            logp_eval = ev[1].reshape(T, len(sample)).float().cpu()
        col.after_rollouts()
    final = env.export_state(sample).cpu().numpy()
    assert rec.calls <= rec.log_a.shape[0]
    traces = rec.traces()
    active = col.active_pid.cpu().numpy()
    # ---- sampled games on the oracle
    ended = 0
    for j, gidx in enumerate(sample):
        o = AutoResetGame(oracle, seed, gidx, blobs0[j])
        gl = GameLists(o, int(active[gidx]), T)
        for r in range(gathers):
            used = gl.gather(traces[r][j])
            assert used == len(traces[r][j]), (gidx, r, used, len(traces[r][j]))      # the device applied nothing the lists did not need
            s = snaps[r]
            assert len(gl.obs) == T + 1 and len(gl.acts) == T and len(gl.rews) >= T, (gidx, r)
            for t in range(T + 1):
                f, lists, lens = gl.obs[t]
                assert np.array_equal(s["obs_f"][t, j].float().numpy(), f), (gidx, r, t)
                assert np.array_equal(s["lens"][t, j].numpy().astype(np.int32), lens), (gidx, r, t)
                for li in range(5):
                    assert np.array_equal(s["lists"][t, j, li, :lens[li]].numpy().astype(np.int32), lists[li, :lens[li]]), (gidx, r, t, li)
                assert float(s["masks"][t, j]) == gl.masks[t], (gidx, r, t)
            for t in range(T):
                assert np.array_equal(s["actions"][t, j].numpy(), gl.acts[t].astype(np.int64)), (gidx, r, t)
                assert np.array_equal(s["am_float"][t, j].numpy(), gl.amasks[t]), (gidx, r, t)
                assert abs(float(s["rewards"][t, j]) - float(np.float32(gl.rews[t]))) < 1e-6, (gidx, r, t, float(s["rewards"][t, j]), gl.rews[t])
            gl.after()
        assert np.array_equal(o.export(), final[j]), gidx
        ended += int(gl.ends > 0)
    assert ended >= 8, f"only {ended} of the {len(sample)} sampled games ended inside the rollouts"
    # ---- the rollout's stored log-probs against the learner's evaluation of the same rows
    gap = (snaps[-1]["action_log_probs"].float() - logp_eval).abs()
    print(f"config 3 at full size: {len(sample)} sampled games, {ended} ended inside a rollout; |logp_rollout - logp_learner| max {float(gap.max()):.4f} "
          f"mean {float(gap.mean()):.5f}; games_complete {complete0}; buckets {col.bucket_log}")
    assert float(gap.max()) < LOGP_ROLLOUT_VS_LEARNER_TOL, float(gap.max())


def test_four_full_size_collector_sets_in_one_process(hip_lib):
    """tools/rollout_schedules.py of round 4 died (SIGSEGV inside hipGraphLaunch) at its FOURTH 65 536-game env + collector + storage in
    one process.  The same sequence - one captured policy pass, the halving buckets, the default buckets twice - each set used for a
    rollout of T = 200 and dropped; device memory returns to where it was.  (The captured passes are single-chain graphs now:
    policy._Branches.in_graphs)"""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy, _Branches
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd import _lib
    rt = int(_lib.lib().catan_hip_runtime_version())
    # the gate (round 6): branched graphs only on a runtime listed as good or by explicit opt-in; this image's runtime is not listed
    assert rt > 0 and _Branches.graphs_allowed() == (rt in _Branches.GOOD_RUNTIMES or os.environ.get("CATAN_GRAPH_BRANCHES") == "1")
    n, T = 65536, 200
    torch.manual_seed(0)
    net = CatanPolicy().cuda()
    torch.cuda.synchronize()
    used = []
    for k, kw in enumerate((dict(act_buckets=(n,)), dict(act_buckets=tuple(n >> j for j in range(5))), dict(), dict(deferred_window=4))):
        env = VecCatanEnv(n, seed=0)
        env.random_rollout(0, 600)
        col = RolloutCollector(env, net, T, seed=0, autocast_dtype=torch.bfloat16, **kw)
        st = col.gather_rollouts()
        assert bool((col.n_obs == T + 1).all()) and env.invalid_action_count() == 0, k
        col.after_rollouts()
        col.close()
        env.close()
        del col, st, env
        gc.collect()
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info()
        used.append((total - free) / 2 ** 30)
    assert max(used) - min(used) < 3.0, used                                # GB: nothing of a set stays behind
