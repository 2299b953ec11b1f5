"""The device collector's two schedules against each other and against the reference's rollouts: catan_step vs
catan_step_deferred (games that wait for the slow path), all games vs only the games that still miss observations (bucketed
policy passes).  The bookkeeping of game_manager.py:78-136 must not see the difference."""
import numpy as np
import pytest
import torch

import rollout_fixture as rf
from settlers_of_catan_rl_amd import spec

pytestmark = pytest.mark.gpu


def _hip_env(n, seed, **kw):
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    return VecCatanEnv(n, seed=seed, **kw)


class SamplerPolicy(object):
    """The library's uniform-random legal policy keyed by (game, number of decisions that game has taken): the action a game
    is given does not depend on the iteration it is evaluated in nor on the row it occupies in the policy pass."""
    include_lstm = False
    wants_games = True

    def __init__(self, cenv):
        self.cenv = cenv

    def act(self, f, lists, lens, masks, generator=None, deterministic=False, games=None, **_kw):
        cnt = self.cenv.steps_taken
        rows = torch.arange(self.cenv.n) if games is None else games.cpu().long()
        c = cnt[rows]
        a = torch.zeros((rows.numel(), spec.ACTION_WORDS), dtype=torch.int64, device=f.device)
        for v in torch.unique(c).tolist():
            allg = self.cenv.env.sample_random_actions(int(v)).long()
            m = (c == v).to(f.device)
            a = torch.where(m[:, None], allg[rows.to(f.device)], a)
        return torch.zeros(a.shape[0], 1, device=f.device), a, rf.scripted_log_prob(a)[:, None].to(f.device)


def _run(n, T, seed, gathers, **ckw):
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    env = _hip_env(n, seed)
    env.random_rollout(0, 150)                       # past the initial placements: roads, settlements, robber, trades
    cenv = rf.CountingEnv(env)
    col = RolloutCollector(cenv, SamplerPolicy(cenv), T, seed=seed, **ckw)
    out = []
    for _ in range(gathers):
        st = col.gather_rollouts()
        snap = {k: getattr(st, k).clone().cpu() for k in ("obs_f", "lists", "lens", "masks", "rewards", "actions", "action_log_probs", "action_masks")}
        snap["games_complete"] = st.games_complete
        snap["state"] = env.export_state().cpu()
        snap["n_obs"] = col.n_obs.clone().cpu(); snap["iters"] = col.iters
        out.append(snap)
        col.after_rollouts()
    assert env.invalid_action_count() == 0
    return out


def _same(a, b, what):
    for g, (x, y) in enumerate(zip(a, b)):
        for k in x:
            if k == "iters":
                continue
            if torch.is_tensor(x[k]):
                assert torch.equal(x[k], y[k]), (what, "gather", g, k, int((x[k] != y[k]).sum()))
            else:
                assert x[k] == y[k], (what, g, k, x[k], y[k])


def test_deferred_and_bucketed_collectors_equal_the_lockstep_one(hip_lib):
    n, T, seed, gathers = 1536, 12, 5, 3
    base = _run(n, T, seed, gathers, deferred_window=0)
    assert sum(int((s["masks"] == 0).sum()) for s in base) >= 0 and base[-1]["games_complete"] >= 0
    for ckw in (dict(), dict(deferred_window=1), dict(deferred_window=0, act_buckets=(n // 8, n // 4, n // 2)),
                dict(deferred_window=8, act_buckets=(n // 16, n // 4))):
        got = _run(n, T, seed, gathers, **ckw)
        _same(base, got, ckw)
        if ckw.get("deferred_window", 4):
            assert got[0]["iters"] >= base[0]["iters"]            # waiting games take part in more iterations


def test_rollout_fixture_on_the_deferred_and_bucketed_collector(hip_lib):
    """tests/golden/rollout_small.npz (the reference manager's own rollouts) through catan_step_deferred and through policy passes
    over game lists: every tensor of every rollout is still the reference's."""
    for ckw in (dict(deferred_window=0), dict(deferred_window=3), dict(deferred_window=2, act_buckets=(1, 2, 4)), dict(deferred_window=0, act_buckets=(2, 3))):
        envs = []
        rf.check_rollout_fixture(lambda n, seed: envs.append(_hip_env(n, seed)) or envs[-1], collector_kwargs=ckw)
        assert envs[0].invalid_action_count() == 0
