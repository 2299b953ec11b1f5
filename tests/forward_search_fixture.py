"""tests/golden/forward_search.npz (tools/gen_golden.py gen_forward_search) against settlers_of_catan_rl_amd.forward_search:
the reference's proposal procedure, UCB bookkeeping and simulator as recorded data - shared by the CPU test (oracle-backed env)
and the `-m gpu` test (HIP env, net on the device)."""
import random

import numpy as np
import torch

import golden_util as gu
import policy_fixture as pf
from settlers_of_catan_rl_amd import forward_search as fs


def fixture_net(device):
    """CatanPolicy with the fixture weights ("ff:" salt - the weights the generator loaded into the reference net)."""
    g = gu.load("policy_small.npz")
    net, _ = pf.load_fixture_policy(g, "ff", device)
    return net


def check_proposals(net, device):
    """default_sample_actions (sample_actions_fn.py:55-329): same proposal lists, root by root (arg-max heads, same random.seed)"""
    g = gu.load("forward_search.npz")
    f = torch.from_numpy(g["prop_obs_f"].astype(np.float32)).to(device)
    lists = torch.from_numpy(g["prop_lists"].astype(np.int32)).to(device); lens = torch.from_numpy(g["prop_lens"].astype(np.int32)).to(device)
    masks = torch.from_numpy(np.unpackbits(g["prop_masks"], axis=1, bitorder="little")[:, :325].astype(np.float32)).to(device)
    rngs = [random.Random(int(s)) for s in g["prop_seed"]]
    got, counts = fs.propose_actions(net, f, lists, lens, masks, 10, initial_settlement_phase=[bool(x) for x in g["prop_initial"]], rngs=rngs,
                                     deterministic=True)
    got = got.cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    off, kinds = 0, set()
    for i, c in enumerate(g["prop_count"]):
        want = g["prop_actions"][off:off + c].astype(np.int64); off += c
        assert int(counts[i]) == int(c), (i, int(counts[i]), int(c))
        assert np.array_equal(got[i, :c], want), (i, got[i, :c], want)
        kinds |= set(want[:, 0].tolist())
    assert len(kinds) >= 6, kinds
    return len(g["prop_count"])


def check_ucb():
    """_select_action / _update_stats / MovingAvgCalculator (policy.py:151-177): every selection, the final choices, the std"""
    g = gu.load("forward_search.npz")
    n_acts, sel, vals = g["ucb_n_act"], g["ucb_sel"], g["ucb_vals"]
    R, K = sel.shape[1], sel.shape[2]
    st = fs.UCBStats(R, 10)
    rounds = sel.shape[0] // n_acts.shape[0]
    for d in range(n_acts.shape[0]):
        st.new_decision(n_acts[d])
        for rnd in range(rounds):
            ids = sel[d * rounds + rnd]
            for k in range(K):
                a = st.select(True); st.start(a)
                assert np.array_equal(np.asarray(a), ids[:, k]), (d, rnd, k, a, ids[:, k])
            for k in range(K):
                st.update(vals[d * rounds + rnd][:, k], ids[:, k])
        assert np.array_equal(np.asarray(st.select(False)), g["ucb_best"][d]), d
        assert np.allclose(np.asarray(st.last_std), g["ucb_std"][d], rtol=0, atol=1e-9), d


def check_simulations(net, make_env, device, rel_tol=2e-3):
    """run_simulation_forward + gae (worker.py:61-143): the value estimate of every recorded start state.  make_env(n, seed)
    -> dense-reward env without auto-reset whose game i draws from the Philox stream (seed, i)."""
    g = gu.load("forward_search.npz")
    worst = 0.0
    for k in range(len(g["sim_value"])):
        env = make_env(1, int(g["sim_seed"][k]))
        env.import_state(g["sim_blob"][k:k + 1])
        got = fs.simulate(env, net, torch.tensor([int(g["sim_ctrl"][k])], device=device), torch.from_numpy(g["sim_init"][k:k + 1].astype(np.int64)).to(device),
                          max_depth=int(g["sim_depth"][k]), gamma=0.999, deterministic=True)
        want = float(g["sim_value"][k])
        err = abs(float(got[0]) - want) / max(1.0, abs(want))
        assert err < rel_tol, (k, float(got[0]), want)
        worst = max(worst, err)
    return worst
