"""Rollout collection bookkeeping (rollout.RolloutCollector) on CPU:
  (1) lock-step vectorised collector == a sequential per-game restatement of RL/ppo/game_manager.py:69-150 (Python lists),
      on late-game states so that game ends, resets and carry-over between rollouts are exercised;
  (2) against the upstream GamesAndPoliciesManager itself (one game, identical net weights, arg-max actions)."""
import os
import sys

import numpy as np
import pytest
import torch

from settlers_of_catan_rl_amd import spec
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.policy import CatanPolicy
from oracle_vec_env import OracleVecEnv, ScriptedPolicy, RecurrentScriptedPolicy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/RL/ppo")


def _sequential_manager(env, policy, active_pid, T, state, L=0):
    """game_manager.py:69-140 for every game of `env`, one game at a time is impossible with a lock-step env object, so the
    restatement keeps per-game Python lists and walks all games step by step - the list logic per game is the reference's."""
    n = env.n
    if state is None:                                      # game_manager.py:35-59
        state = dict(obs=[[] for _ in range(n)], masks=[[1.0] for _ in range(n)], acts=[[] for _ in range(n)],
                     lps=[[] for _ in range(n)], rews=[[] for _ in range(n)], hid=[[] for _ in range(n)],
                     cur=[{p: (torch.zeros(L), torch.zeros(L)) for p in (1, 2, 3, 4)} for _ in range(n)])
        f, lists, lens = env.get_obs(); dec = env.deciding_player()
        for i in range(n):
            if int(dec[i]) == active_pid[i]:
                state["obs"][i].append((f[i].clone(), lists[i].clone(), lens[i].clone()))
                state["hid"][i].append(state["cur"][i][active_pid[i]])
    racc = np.zeros((n, 4)); done_since = [False] * n
    term = [state["masks"][i][0] for i in range(n)]                  # game_manager.py:74-75
    while any(len(state["obs"][i]) < T + 1 for i in range(n)):
        dec = env.deciding_player(); f, lists, lens = env.get_obs(); masks = env.get_action_masks()
        frozen = [len(state["obs"][i]) >= T + 1 for i in range(n)]
        if L:
            h = torch.stack([state["cur"][i][int(dec[i])][0] for i in range(n)])
            c = torch.stack([state["cur"][i][int(dec[i])][1] for i in range(n)])
            _, a, lp, (nh, nc) = policy.act(f, lists, lens, masks, hidden=(h, c), nonterminal=torch.tensor(term))
            for i in range(n):
                if not frozen[i]:
                    state["cur"][i][int(dec[i])] = (nh[i].clone(), nc[i].clone())       # :89
        else:
            _, a, lp = policy.act(f, lists, lens, masks)
        a_env = a.to(torch.int32)
        for i in range(n):
            if frozen[i]:
                a_env[i, 0] = -1
        rew, done = env.step(a_env)
        ndec = env.deciding_player(); nf, nlists, nlens = env.get_obs()
        for i in range(n):
            if frozen[i]:
                continue
            act = active_pid[i]
            racc[i] += rew[i].numpy()
            d = bool(done[i])
            term[i] = 1.0 - float(d)                                 # :97
            reward_updated = False
            if int(dec[i]) == act:
                state["acts"][i].append(a[i].clone()); state["lps"][i].append(float(lp[i, 0]))
            # NB the reference evaluates `n_players_go` BEFORE the reset here; for a finished game exactly one reward is
            # appended either way (lines 106-118), which is what is restated
            if d:
                state["rews"][i].append(racc[i, act - 1]); racc[i, act - 1] = 0.0; reward_updated = True
            elif int(ndec[i]) == act and len(state["acts"][i]) > 0 and not done_since[i]:
                state["rews"][i].append(racc[i, act - 1]); racc[i, act - 1] = 0.0; reward_updated = True
            if d:
                state["masks"][i].append(0.0); done_since[i] = False; racc[i] = 0.0
                state["cur"][i] = {p: (torch.zeros(L), torch.zeros(L)) for p in (1, 2, 3, 4)}    # :121-124
            if int(ndec[i]) == act:
                if not d and not done_since[i]:
                    state["masks"][i].append(1.0)
                done_since[i] = False
                state["obs"][i].append((nf[i].clone(), nlists[i].clone(), nlens[i].clone()))
                state["hid"][i].append(state["cur"][i][act])                                     # :133
            elif d:
                done_since[i] = True
    return state


def _after(state):                                        # game_manager.py:142-150
    for k in ("acts", "lps", "rews"):
        state[k] = [[] for _ in state[k]]
    state["obs"] = [[o[-1]] for o in state["obs"]]
    state["hid"] = [[h[-1]] for h in state["hid"]]
    state["masks"] = [[m[-1]] for m in state["masks"]]


@pytest.mark.parametrize("recurrent", [False, True])
def test_collector_matches_sequential_restatement(recurrent):
    n, T, seed = 20, 24, 13
    envA, envB = OracleVecEnv(n, seed), OracleVecEnv(n, seed)
    envA.advance_random(1750); envB.advance_random(1750)             # late game: several games end inside the rollouts
    Pol = RecurrentScriptedPolicy if recurrent else ScriptedPolicy
    L = Pol.lstm_size if recurrent else 0
    col = RolloutCollector(envA, Pol(envA), T, seed=4)
    active = [int(x) for x in col.active_pid]
    state = None
    polB = Pol(envB)
    ends = 0
    for r in range(3):
        st = col.gather_rollouts()
        state = _sequential_manager(envB, polB, active, T, state, L)
        for i in range(n):
            assert len(state["obs"][i]) == T + 1 and len(state["acts"][i]) == T
            for t in range(T + 1):
                assert torch.equal(st.obs_f[t, i], state["obs"][i][t][0]), (r, i, t)
                assert torch.equal(st.lists[t, i].int(), state["obs"][i][t][1]) and torch.equal(st.lens[t, i].int(), state["obs"][i][t][2])
                assert float(st.masks[t, i]) == state["masks"][i][t], (r, i, t)
                if recurrent:
                    assert torch.equal(st.hidden[0, t, i], state["hid"][i][t][0]) and torch.equal(st.hidden[1, t, i], state["hid"][i][t][1]), (r, i, t)
            for t in range(T):
                assert torch.equal(st.actions[t, i], state["acts"][i][t])
                assert float(st.action_log_probs[t, i]) == state["lps"][i][t]
                assert abs(float(st.rewards[t, i]) - state["rews"][i][t]) < 1e-6, (r, i, t)
            ends += sum(1 for m in state["masks"][i][:T + 1] if m == 0.0)
        # the stored action masks are the masks the deciding seat saw
        col.after_rollouts(); _after(state)
    assert ends >= 3, "the test is meant to cover game ends"
    assert np.array_equal(envA.b.export(), envB.b.export())


def test_action_mask_packing_roundtrip():
    from settlers_of_catan_rl_amd.rollout import pack_action_masks, RolloutStorage
    m = (torch.rand(37, 325) > 0.5).float()
    st = RolloutStorage(1, 1, "cpu")
    assert torch.equal(st.unpack_action_masks(pack_action_masks(m)), m)


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
@pytest.mark.parametrize("lstm", [False, True])
def test_collector_vs_reference_manager(lstm):
    """lstm=True: `include_lstm` of build_agent_model.py:26 switched on - the per-seat LSTM states, their reset at a game
    end and the stored `active_hidden_states` (game_manager.py:54-59,81-89,121-124,133)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_harness as rh
    from RL.ppo.game_manager import GamesAndPoliciesManager
    import RL.models.build_agent_model as bam
    T, seed = (40, 23) if lstm else (14, 21)
    torch.manual_seed(3)
    stream = rh.PhiloxStream(seed, 0)
    old_flag = bam.include_lstm
    bam.include_lstm = lstm
    try:
        with rh.patched_rng(rh.PhiloxStream(seed ^ 0xABC, 0)):         # constructor draws are discarded (as in RefEnv)
            mgr = GamesAndPoliciesManager(num_envs=1, num_steps=T)
    finally:
        bam.include_lstm = old_flag
    sd = mgr.policies[0].state_dict()
    g = torch.Generator().manual_seed(9)
    sd = {k: (v + 0.05 * torch.randn(v.shape, generator=g) if v.numel() and v.dtype == torch.float32 else v) for k, v in sd.items()}
    for p in mgr.policies:
        p.load_state_dict(sd)
        orig = p.act
        p.act = (lambda o: (lambda *a, **kw: o(*a, **{**kw, "deterministic": True})))(orig)      # arg-max actions
    with rh.patched_rng(stream):
        mgr.reset()                                                     # env.reset() under the game's philox stream
        ref = mgr.gather_rollouts()
    # mine: oracle-backed env with the same stream, same weights, same active seat
    env = OracleVecEnv(1, seed)

    class Det(object):
        def __init__(self, net): self.net = net
        def act(self, *a, **kw):
            kw.pop("generator", None)
            return self.net.act(*a, deterministic=True, **kw)
        include_lstm, lstm_size = lstm, 256
    net = CatanPolicy(include_lstm=lstm); net.load_reference_state_dict(sd); net.eval()
    col = RolloutCollector(env, Det(net), T, seed=0)
    col.active_pid[:] = int(mgr.active_player_ids[0])
    col.reset()
    st = col.gather_rollouts()
    obs_ref, hid_ref, rew_ref, act_ref, amask_ref, lp_ref, tm_ref = ref
    if lstm:
        for t in range(T + 1):
            for j in range(2):
                assert torch.allclose(st.hidden[j, t, 0], hid_ref[0][t][j][0], atol=2e-5), (t, j)
        assert float(st.hidden[:, 1:].abs().max()) > 0.01
    o = spec.OBS_FLOAT_OFFSETS
    for t in range(T + 1):
        ro = obs_ref[0][t]
        for k, shp in spec.OBS_FLOAT_KEYS.items():
            got = st.obs_f[t, 0, o[k]:o[k] + int(np.prod(shp))]
            assert torch.equal(got, ro[k].reshape(-1).float()), (t, k)
        for li, k in enumerate(spec.OBS_LIST_KEYS):
            ln = int(st.lens[t, 0, li])
            assert torch.equal(st.lists[t, 0, li, :ln].long(), ro[k][0].reshape(-1)), (t, k)
        assert float(st.masks[t, 0]) == float(tm_ref[0][t])
    for t in range(T):
        # the reference's stored actions were converted in place by torch_act_to_np (RL/models/policy.py:192-199)
        flat = torch.tensor(np.concatenate([np.asarray(h).reshape(-1) for h in act_ref[0][t]]), dtype=torch.int64)
        assert torch.equal(st.actions[t, 0], flat), (t, st.actions[t, 0], flat)
        assert abs(float(st.action_log_probs[t, 0]) - float(lp_ref[0][t])) < 1e-5
        assert abs(float(st.rewards[t, 0]) - float(rew_ref[0][t])) < 1e-6
        mref = torch.cat([(m.transpose(0, 1) if i in (1, 6, 9) else m).reshape(-1) for i, m in enumerate(amask_ref[0][t])])
        assert torch.equal(st.unpack_action_masks(st.action_masks[t, 0]), mref)
