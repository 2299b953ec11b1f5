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


def test_validate_cases_vs_reference(hip_lib):
    """SURVEY a4 on the device - validate mode against the REFERENCE's own verdicts (tests/golden/validate_cases.npz, generated
    by tools/gen_golden.py from `_translate_action` + `Game.validate_action`, game/game.py:264-525): 8 800 probe actions in 720
    reference states; the HIP step must accept exactly what the reference accepts - incl. the ~500 actions NO mask offers
    (MoveRobber onto an empty tile, ProposeTrade past the limit, RollDice during road building, the dummy edge, ...) - leave
    rejected games untouched, and for every accepted probe end in the reference's state / masks / rewards / done."""
    import torch
    g = {k: v for k, v in gu.load("validate_cases.npz").items()}
    acc, inm = g["case_accept"].astype(bool), g["case_in_masks"].astype(bool)
    assert len(acc) > 8000 and (acc & ~inm).sum() > 400
    blob_at = {int(c): k for k, c in enumerate(g["post_blob_case"])}
    st_of = g["case_state"]
    groups = {}
    for c in range(len(acc)):
        si = int(st_of[c])
        groups.setdefault((int(g["state_trades"][si]), int(g["state_max_actions"][si]), int(g["state_seed"][si]), int(g["state_env"][si])), []).append(c)
    assert len(groups) == 4
    checked_oom = 0
    for (tr, ma, seed, env_id), cases in groups.items():
        kw = dict(max_proposed_trades_per_turn=None if tr < 0 else tr, max_actions_per_turn=None if ma < 0 else ma, auto_reset=False)
        cases = np.array(cases)
        draws = acc[cases] & np.isin(g["case_action"][cases, 0], (9, 11))          # RollDice / StealResource draw from the game's stream
        # (1) everything that draws nothing: one game per case
        batch = cases[~draws]
        env = _env(len(batch), seed, env_id0=env_id, **kw)
        r64 = env.enable_reward64()
        before = g["states"][st_of[batch]].astype(np.int32)
        env.import_state(before)
        rew, done = env.step(torch.from_numpy(g["case_action"][batch].astype(np.int32)))
        # (a NEGATIVE action type is the C ABI's explicit no-op - a frozen game of a rollout - and is not counted as an error)
        assert env.invalid_action_count() == int((~acc[batch] & (g["case_action"][batch, 0] >= 0)).sum())
        after = env.export_state().cpu().numpy()
        masks = env.get_action_masks().cpu().numpy()
        decide = env.deciding_player().cpu().numpy()
        rew64, done = r64.cpu().numpy(), done.cpu().numpy().astype(bool)
        for j, c in enumerate(batch):
            a = g["case_action"][c].tolist()
            if not acc[c]:
                assert np.array_equal(after[j], before[j]), f"case {c} {a}: a rejected action changed the game\n" + spec.describe_state_diff(before[j], after[j])
                assert not rew64[j].any() and not done[j], c
                continue
            if c in blob_at:
                want = g["post_blobs"][blob_at[c]].astype(np.int32)
                assert np.array_equal(after[j], want), f"case {c} {a}:\n" + spec.describe_state_diff(want, after[j])
                checked_oom += 1
            assert gu.crc(after[j]) == int(g["post_crc"][c]), (c, a)
            assert np.array_equal(masks[j], gu.unpack_masks(g["post_masks"][c])), (c, a)
            assert np.array_equal(rew64[j], g["post_reward64"][c]) and done[j] == bool(g["post_done"][c]), (c, a)
            assert int(decide[j]) == int(g["post_deciding"][c]), (c, a)
        # (2) the accepted dice rolls and steals, on the game whose Philox stream the reference used
        one = _env(1, seed, env_id0=env_id, **kw)
        r64 = one.enable_reward64()
        for c in cases[draws]:
            one.import_state(g["states"][st_of[c]].astype(np.int32)[None])
            _, done = one.step(torch.from_numpy(g["case_action"][c].astype(np.int32))[None])
            post = one.export_state()[0].cpu().numpy()
            if c in blob_at:
                want = g["post_blobs"][blob_at[c]].astype(np.int32)
                assert np.array_equal(post, want), f"case {c}:\n" + spec.describe_state_diff(want, post)
                checked_oom += 1
            assert gu.crc(post) == int(g["post_crc"][c]), c
            assert np.array_equal(one.get_action_masks()[0].cpu().numpy(), gu.unpack_masks(g["post_masks"][c])), c
            assert np.array_equal(r64[0].cpu().numpy(), g["post_reward64"][c]) and bool(done[0].item()) == bool(g["post_done"][c]), c
        assert one.invalid_action_count() == 0
    assert checked_oom > 400


def test_mt19937_known_answer_on_the_hip_path(hip_lib):
    """RNG contract (A) on the DEVICE (VERDICT r5 missing #2): tests/golden/mt_kat.npz holds trajectories of the UNPATCHED
    reference - `np.random.seed(s); random.seed(s); env = EnvWrapper(); env.reset()`, then random legal actions (env/wrapper.py:30-50) - as
    per-step state CRCs.  The HIP path replays them draw for draw: the two MT19937 generators live in device memory
    (catan_seed_mt19937 / catan_mt19937_set_state: numpy's init_genrand + masked rejection for np.random.shuffle / randint, CPython's
    init_by_array + top-bits getrandbits for the steal's random.choice), the board / game / wrapper resets take their draws in the
    reference's order (Board(), Game(), EnvWrapper.reset()).  Once through env.EnvWrapper picking the generators up from the process's
    np.random / random state, once through the C ABI's own seeding."""
    import random
    import torch
    from settlers_of_catan_rl_amd.env import EnvWrapper, VecCatanEnv
    g = gu.load("mt_kat.npz")
    steps = resets = steals = 0
    for s in g["seeds"]:
        s = int(s)
        acts, crcs = g[f"actions_{s}"], g[f"crc_{s}"]
        for form in ("process globals", "catan_seed_mt19937"):
            if form == "process globals":
                np.random.seed(s); random.seed(s)
                before = np.random.get_state()[1].copy()
                env = EnvWrapper(rng="mt19937")
                assert np.array_equal(np.random.get_state()[1], before)          # copies: the process's own generators stay put
                env.reset()
                vec = env.vec
            else:
                vec = VecCatanEnv(1, seed=123, auto_reset=False)
                vec.seed_mt19937(s, s)
                vec.reset_board_only(); vec.reset(); vec.reset()
            for t in range(len(acts)):
                b = vec.export_state()[0].cpu().numpy(); b[-1] = 0
                assert gu.crc(b) == int(crcs[t]), (s, form, t)
                _, done = vec.step(torch.from_numpy(acts[t].astype(np.int32)).view(1, -1))
                steps += 1
                steals += int(acts[t][0] == 11)
                if bool(done[0].item()):
                    vec.reset(); resets += 1
            b = vec.export_state()[0].cpu().numpy(); b[-1] = 0
            assert np.array_equal(b, g[f"final_{s}"]), (s, form)
            assert vec.invalid_action_count() == 0
    assert steps > 1000 and steals > 0, (steps, resets, steals)
    # the contract is for one game, lock-step: everything else refuses
    big = VecCatanEnv(4, seed=0)
    with pytest.raises(Exception, match="single-game"):
        big.seed_mt19937(1, 1)
    with pytest.raises(Exception, match="MT19937"):
        vec.random_rollout_deferred(8, 4)

