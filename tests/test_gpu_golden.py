"""GPU vs golden vectors captured from the upstream reference itself (tools/gen_golden.py): the HIP path is
checked directly against reference outputs, not only against the oracle.  Bit-exact."""
import numpy as np
import pytest

import golden_util as gu
from settlers_of_catan_rl_amd import spec

pytestmark = pytest.mark.gpu


def _env(n, seed, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    return VecCatanEnv(n, seed=seed, **kw)


def test_reset_states_vs_reference(hip_lib):
    g = gu.load("reset_states.npz")
    env = _env(len(g["blobs"]), int(g["seed"]))
    got = env.export_state().cpu().numpy()
    assert np.array_equal(got, g["blobs"]), spec.describe_state_diff(g["blobs"][0], got[0])


@pytest.mark.parametrize("name", gu.TRAJS)
def test_trajectory_vs_reference(hip_lib, name):
    import torch
    t = gu.load(name)
    dense, anneal, trades, max_actions = gu.traj_kwargs(t)
    env = _env(1, int(t["seed"]), env_id0=int(t["env_id"]), auto_reset=True, dense_reward=dense, max_proposed_trades_per_turn=trades,
               max_actions_per_turn=max_actions)
    env.set_reward_annealing_factor(anneal)
    r64 = env.enable_reward64()
    sample = {int(i): k for k, i in enumerate(t["sample_idx"])}
    n = len(t["actions"])
    for step in range(n):
        blob = env.export_state()[0].cpu().numpy()
        assert gu.crc(blob) == int(t["state_crc"][step]), f"state crc differs at step {step}"
        m = env.get_action_masks()[0].cpu().numpy()
        assert np.array_equal(m, gu.unpack_masks(t["masks"][step])), f"masks differ at step {step}"
        if step in sample:
            assert int(env.deciding_player()[0].item()) == int(t["deciding"][step])
            assert np.array_equal(blob, t["sample_blob"][sample[step]].astype(np.int32))
        a = torch.from_numpy(t["actions"][step].astype(np.int32)).view(1, spec.ACTION_WORDS)
        rew, done = env.step(a)
        assert np.array_equal(rew[0].cpu().numpy(), t["rewards"][step]) and bool(done[0].item()) == bool(t["dones"][step]), step
        if "rewards64" in t.files:      # the reference's Python-float rewards, before the single rounding to fp32
            assert np.array_equal(r64[0].cpu().numpy(), t["rewards64"][step]), step
    assert env.invalid_action_count() == 0
    assert np.array_equal(env.export_state()[0].cpu().numpy(), t["final_blob"])


def test_reference_states_masks(hip_lib):
    """Import every sampled reference state of every golden trajectory into the device and compare the masks (one env per
    trade limit the trajectories were generated with: the propose-trade mask bit depends on it, wrapper.py:284-289)."""
    groups = {}
    for name in gu.TRAJS:
        t = gu.load(name)
        _, _, trades, max_actions = gu.traj_kwargs(t)
        blobs, masks = groups.setdefault((trades, max_actions), ([], []))
        for k, i in enumerate(t["sample_idx"]):
            blobs.append(t["sample_blob"][k].astype(np.int32)); masks.append(gu.unpack_masks(t["masks"][int(i)]))
    assert len(groups) >= 4
    for (trades, max_actions), (blobs, masks) in groups.items():
        env = _env(len(blobs), 0, max_proposed_trades_per_turn=trades, max_actions_per_turn=max_actions)
        env.import_state(np.array(blobs))
        assert np.array_equal(env.export_state().cpu().numpy(), np.array(blobs))
        assert np.array_equal(env.get_action_masks().cpu().numpy(), np.array(masks)), trades


def test_randomise_uncertainty_golden_and_oracle(oracle, hip_lib):
    """k_randomise_uncertainty against the reference's before/after states (tests/golden/randomise.npz) and, on a larger
    batch of mid-game states, against the CPU oracle (state incl. card orders, hands, pile, RNG draw count; masks)."""
    import torch
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    g = gu.load("randomise.npz")
    seed = int(g["seed"])
    n_games = int(g["env_id"].max()) + 1
    per = len(g["ctrl"]) // n_games
    env = VecCatanEnv(n_games, seed=seed, auto_reset=False)
    for k in range(per):                                    # one case per game per round: game e uses the stream of env id e
        sel = [e * per + k for e in range(n_games)]
        assert [int(g["env_id"][i]) for i in sel] == list(range(n_games))
        env.import_state(g["before"][sel])
        env.randomise_uncertainty(torch.tensor(g["ctrl"][sel]))
        out = env.export_state().cpu().numpy()
        for j, i in enumerate(sel):
            assert np.array_equal(out[j], g["after"][i]), f"case {i}:\n" + spec.describe_state_diff(g["after"][i], out[j])
    assert env.inconsistent_deal_count() == 0
    # larger batch vs the oracle: 512 games at various ages, a random controlling player each (0 = untouched)
    n, seed = 512, 77
    env = VecCatanEnv(n, seed=seed)
    ob = oracle.OracleBatch(n, seed)
    env.random_rollout(0, 900)
    ob.run_random(900, want_blobs=False)
    rng = np.random.default_rng(5)
    ctrl = rng.integers(0, 5, size=n)
    env.randomise_uncertainty(torch.tensor(ctrl))
    for i in range(n):
        if ctrl[i]:
            ob.L.orc_randomise_uncertainty(ob.env_ptr(i), int(ctrl[i]))
    out = env.export_state().cpu().numpy()
    want = ob.export()
    bad = np.flatnonzero((out != want).any(axis=1))
    assert len(bad) == 0, f"{len(bad)} games differ; game {bad[0]} ctrl {ctrl[bad[0]]}:\n" + spec.describe_state_diff(want[bad[0]], out[bad[0]])
    assert np.array_equal(env.get_action_masks().cpu().numpy(), ob.masks())
    assert env.inconsistent_deal_count() == 0
