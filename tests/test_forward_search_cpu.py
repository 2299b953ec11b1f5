"""Forward search (reference RL/forward_search_policy/*) on CPU: every deterministic piece against the imported reference
(development container; skipped where /root/reference is absent), the rest against restated formulas."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_bootstrap  # noqa: E402

from settlers_of_catan_rl_amd import forward_search as fs, spec  # noqa: E402
from settlers_of_catan_rl_amd.policy import CatanPolicy  # noqa: E402
from oracle_vec_env import OracleVecEnv  # noqa: E402

HAVE_REF = ref_bootstrap.have_reference()
if HAVE_REF:
    ref_bootstrap.bootstrap()


def _ref_gae(values, rewards, gamma, done):
    """worker.gae restated literally (used when the reference is not mounted)."""
    lam = 0.95
    if len(values) <= 1:
        return rewards[0] + gamma * rewards[1]
    first, rg, vg = rewards[0], rewards[1:-1], values[:-1]
    ns = len(vg) - 1
    g = 0.0
    for step in reversed(range(ns)):
        delta = rg[step] + gamma * vg[step + 1] - vg[step]
        g = delta if (step == ns - 1 and done) else delta + gamma * lam * g
    return first + gamma * (g + vg[0])


def test_gae_estimate_matches_worker_gae():
    gae = _ref_gae
    if HAVE_REF:
        from RL.forward_search_policy.worker import gae
    rng = np.random.default_rng(0)
    n, D = 200, 20
    values = rng.normal(100, 40, size=(n, D + 1)); rewards = rng.normal(2, 5, size=(n, D + 2))
    nv = rng.integers(0, D, size=n); done = rng.random(n) < 0.4
    nr = np.where(nv <= 1, rng.integers(2, 4, size=n), nv + 1 + done.astype(int))
    got = fs.gae_estimate(values, nv, rewards, nr, 0.999, done)
    for i in range(n):
        want = gae([float(x) for x in values[i, :nv[i]]], [float(x) for x in rewards[i, :nr[i]]], gamma=0.999, done=bool(done[i]))
        assert abs(got[i] - float(want)) < 1e-9 * max(1.0, abs(float(want))), i


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_ucb_stats_match_reference_policy_object():
    """_select_action / _update_stats / MovingAvgCalculator: drive a reference ForwardSearchPolicy shell (no worker
    processes) and UCBStats with the same stream of simulation results."""
    from RL.forward_search_policy.policy import ForwardSearchPolicy
    from RL.forward_search_policy.utils import MovingAvgCalculator
    rng = np.random.default_rng(1)
    R, A = 3, 10
    refs = []
    for r in range(R):
        o = ForwardSearchPolicy.__new__(ForwardSearchPolicy)
        o.value_moving_average = MovingAvgCalculator(window_size=500)
        refs.append(o)
    st = fs.UCBStats(R, A)
    for decision in range(3):                              # the moving average persists across decisions
        n_act = rng.integers(2, A + 1, size=R)
        st.new_decision(n_act)
        for r, o in enumerate(refs):
            o.proposed_actions = list(range(n_act[r]))
            o.num_simulations_finished = 0; o.num_simulations_in_progress = 0
            o.num_simulations_finished_each_action = np.zeros(n_act[r]); o.num_simulations_started_each_action = np.zeros(n_act[r])
            o.exploit_scores = np.zeros(n_act[r])
        for rnd in range(40):
            K = 4
            ids = np.zeros((R, K), dtype=np.int64)
            for k in range(K):
                a = st.select(True); st.start(a); ids[:, k] = a
                for r, o in enumerate(refs):
                    ar = o._select_action()
                    assert ar == a[r], (decision, rnd, k, r)
                    o.num_simulations_in_progress += 1; o.num_simulations_started_each_action[ar] += 1
            vals = rng.normal(120, 60, size=(R, K)) + 10 * ids
            for k in range(K):
                st.update(vals[:, k], ids[:, k])
                for r, o in enumerate(refs):
                    o._update_stats(vals[r, k], ids[r, k])
        best = st.select(False)
        for r, o in enumerate(refs):
            assert o._select_action(explore=False) == best[r]
            assert abs(o.value_moving_average.get_std() - st.last_std[r]) < 1e-9


def _perturbed_reference_net(seed=9, lstm=False):
    import RL.models.build_agent_model as bam
    old_flag = bam.include_lstm
    bam.include_lstm = lstm
    try:
        net = bam.build_agent_model(device="cpu")
    finally:
        bam.include_lstm = old_flag
    g = torch.Generator().manual_seed(seed)
    sd = {k: (v + 0.05 * torch.randn(v.shape, generator=g) if v.numel() and v.dtype == torch.float32 else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net.eval()
    orig = net.act
    net.act = lambda *a, **kw: orig(*a, **{**kw, "deterministic": True})      # arg-max decisions on both sides
    mine = CatanPolicy(include_lstm=lstm); mine.load_reference_state_dict(sd); mine.eval()
    return net, mine


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
@pytest.mark.parametrize("lstm", [False, True])
def test_simulate_matches_run_simulation_forward(lstm):
    """The batched simulator against the reference's run_simulation_forward: same start state, same (arg-max) policy for
    every seat, same philox game stream; the value estimate must agree (fp32 net on both sides).  lstm: the LSTM policy with
    a different (h, c) per seat at the start and the searching seat's post-decision state (worker.py:61-95)."""
    import ref_harness as rh
    from RL.forward_search_policy.worker import run_simulation_forward
    ref_net, net = _perturbed_reference_net(lstm=lstm)
    gh = torch.Generator().manual_seed(77)
    checked = 0
    for (seed, warm, depth) in [(5, 40, 6), (5, 400, 8), (8, 900, 5), (11, 1500, 20)]:
        rng = np.random.default_rng(seed)
        ref = rh.RefEnv(seed, 0, dense_reward=True)
        obs = ref.reset()
        env = OracleVecEnv(1, seed)
        # drive both to the same mid-game state with the same random legal actions
        import ctypes as C
        for s in range(warm):
            a = rh.random_legal_action(ref.masks(), ref.env, rng)
            obs, _, done = ref.step(a)
            r = np.zeros(4, dtype=np.float32); d = C.c_int(0); ai = np.ascontiguousarray(a, dtype=np.int32)
            env.L.orc_step(env.b.env_ptr(0), ai.ctypes.data_as(C.POINTER(C.c_int32)), r.ctypes.data_as(C.POINTER(C.c_float)), C.byref(d))
            assert bool(d.value) == done
            if done:
                obs = ref.reset(); env.L.orc_game_reset(env.b.env_ptr(0))
        assert np.array_equal(ref.state_blob(), env.b.export()[0])
        ctrl = ref.deciding_player()
        init = rh.random_legal_action(ref.masks(), ref.env, rng)
        # the oracle env must pay dense rewards like EnvWrapper(dense_reward=True)
        blob = env.b.export()
        env_dense = OracleVecEnv(1, seed, dense_reward=True, auto_reset=False)
        env_dense.import_state(blob)
        hid = 0.3 * torch.randn(2, 1, 4, 256, generator=gh) if lstm else None      # every seat's (h, c) at the start state
        init_h = 0.3 * torch.randn(2, 1, 256, generator=gh) if lstm else None       # the searching seat's state after its decision
        with rh.patched_rng(ref.stream):
            want = run_simulation_forward(ref.env, ref_net, player_id=rh.PIDS[ctrl - 1], init_action=rh.action_to_heads(init),
                                          init_player_hs=(init_h[0], init_h[1]) if lstm else None,
                                          curr_hidden_states={p: ((hid[0, :, int(p) - 1], hid[1, :, int(p) - 1]) if lstm else None) for p in rh.PIDS},
                                          curr_obs=ref_net.obs_to_torch(copy.deepcopy(obs)), max_depth=depth, gamma=0.999)
        got = fs.simulate(env_dense, net, torch.tensor([ctrl]), torch.tensor(np.asarray(init)[None]), max_depth=depth, gamma=0.999,
                          deterministic=True, hidden=hid, init_hidden=init_h)
        assert abs(float(got[0]) - float(want)) < 2e-3 * max(1.0, abs(float(want))), (seed, warm, got, want)
        checked += 1
    assert checked == 4


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_propose_actions_matches_default_sample_actions():
    """The batched proposal procedure against default_sample_actions root by root: arg-max policy heads on both sides, the
    procedure's random.choice calls seeded identically."""
    import random
    import ctypes as C
    import ref_harness as rh
    from RL.forward_search_policy.sample_actions_fn import default_sample_actions
    ref_net, net = _perturbed_reference_net(seed=4)
    cases = []
    for (seed, warm) in [(3, 0), (3, 5), (3, 60), (3, 300), (6, 700), (6, 1100), (9, 1500), (9, 2100)]:
        rng = np.random.default_rng(seed + warm)
        ref = rh.RefEnv(seed, 0)
        obs = ref.reset()
        for s in range(warm):
            obs, _, done = ref.step(rh.random_legal_action(ref.masks(), ref.env, rng))
            if done:
                obs = ref.reset()
        random.seed(1234 + warm)
        masks_t = ref_net.act_masks_to_torch(ref.env.get_action_masks())
        initial = bool(ref.env.game.initial_placement_phase)
        want, _ = default_sample_actions(ref_net.obs_to_torch(copy.deepcopy(obs)), None, masks_t, ref_net, 10,
                                         initial_settlement_phase=initial)
        want = np.array([np.concatenate([np.asarray(h).reshape(-1) for h in a]) for a in want], dtype=np.int64)
        f, lists, lens, _ = rh.obs_flat(obs)
        cases.append((f, lists, lens, rh.masks_flat(ref.masks()), initial, 1234 + warm, want))
    f = torch.tensor(np.stack([c[0] for c in cases])); lists = torch.tensor(np.stack([c[1] for c in cases]))
    lens = torch.tensor(np.stack([c[2] for c in cases])); masks = torch.tensor(np.stack([c[3] for c in cases]))
    rngs = [random.Random(c[5]) for c in cases]
    got, counts = fs.propose_actions(net, f, lists, lens, masks, 10, initial_settlement_phase=[c[4] for c in cases], rngs=rngs,
                                     deterministic=True)
    kinds = set()
    for i, c in enumerate(cases):
        want = c[6]
        assert counts[i] == len(want), (i, counts[i], len(want))
        assert np.array_equal(got[i, :counts[i]], want), (i, got[i, :counts[i]], want)
        kinds |= set(want[:, 0].tolist())
    assert len(kinds) >= 6, kinds                        # the cases exercise a spread of action types


def test_forward_search_end_to_end_on_oracle_env():
    """The planner loop (propose -> UCB rounds of state broadcast + randomise_uncertainty + simulations -> pick) on the
    oracle-backed stand-in env: legal choices, simulation bookkeeping, root states untouched."""
    torch.manual_seed(0)
    R, S, K = 3, 4, 2
    root = OracleVecEnv(R, seed=13)
    root.advance_random(500)
    before = root.b.export().copy()
    net = CatanPolicy().eval()
    search = fs.ForwardSearch(net, lambda n: OracleVecEnv(n, seed=99, dense_reward=True, auto_reset=False), R, max_depth=3,
                              sims_per_root=S, sims_per_round=K)
    chosen, info = search.act(root, deterministic=True)
    assert np.array_equal(root.b.export(), before)                       # the roots are only read
    assert search.sims_run == R * S
    assert (info["finished_each"].sum(1) == S).all()
    import ctypes as C
    for r in range(R):
        a = np.ascontiguousarray(chosen[r], dtype=np.int32)
        assert root.L.orc_action_is_legal(root.b.env_ptr(r), a.ctypes.data_as(C.POINTER(C.c_int32)))
        assert 1 <= info["n_proposed"][r] <= 10


def test_forward_search_lstm_states_and_time_budget():
    """LSTM policy: per-seat states go through proposals and simulations, `next_hidden` is the searching seat's state after
    one LSTM step on its root observation; and the wall-clock budget (policy.py:91,111) instead of a simulation count."""
    torch.manual_seed(1)
    R, K = 2, 2
    root = OracleVecEnv(R, seed=17)
    root.advance_random(300)
    net = CatanPolicy(include_lstm=True).eval()
    hidden = 0.2 * torch.randn(2, R, 4, net.lstm_size)
    search = fs.ForwardSearch(net, lambda n: OracleVecEnv(n, seed=5, dense_reward=True, auto_reset=False), R, max_depth=2,
                              sims_per_root=2, sims_per_round=K)
    chosen, info = search.act(root, deterministic=True, hidden=hidden, zero_opponent_hidden_states=True)
    ctrl = root.deciding_player().long()
    f, lists, lens = root.get_obs()
    ar = torch.arange(R)
    with torch.no_grad():
        _, _, _, (h1, c1) = net.act(f, lists, lens.long(), root.get_action_masks(), deterministic=True,
                                    hidden=(hidden[0, ar, ctrl - 1], hidden[1, ar, ctrl - 1]), nonterminal=torch.ones(R))
    assert torch.allclose(info["next_hidden"][0], h1, atol=1e-6) and torch.allclose(info["next_hidden"][1], c1, atol=1e-6)
    assert search.sims_run == R * 2
    # wall-clock budget: at least one round for every root that has more than one proposal, then it stops
    search.sims_run = 0
    chosen, info = search.act(root, deterministic=True, hidden=hidden, max_thinking_time=0.05)
    assert search.sims_run >= K * int((info["n_proposed"] > 0).sum()) and (info["finished_each"].sum(1) >= K).all()


def test_forward_search_fixture_from_the_reference(oracle):
    """tests/golden/forward_search.npz: the reference's proposals, UCB selections and simulator values as data (the same
    fixture runs on the HIP env + device net under -m gpu)."""
    import forward_search_fixture as ff
    net = ff.fixture_net("cpu")
    assert ff.check_proposals(net, "cpu") == 10
    ff.check_ucb()
    ff.check_simulations(net, lambda n, seed: OracleVecEnv(n, seed, dense_reward=True, auto_reset=False), "cpu")
