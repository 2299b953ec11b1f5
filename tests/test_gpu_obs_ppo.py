"""GPU parity: observation encoder, longest road entry, GAE and PPO-loss kernels - against the CPU oracle and against
golden vectors computed by the upstream reference's own code (tools/gen_golden.py)."""
import ctypes as C

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


def _oracle_obs(ob, n):
    f = np.zeros((n, 1787), dtype=np.float32); lists = np.zeros((n, 5, 25), dtype=np.int32)
    lens = np.zeros((n, 5), dtype=np.int32); pid = np.zeros((n,), dtype=np.int32)
    for i in range(n):
        ob.L.orc_obs(ob.env_ptr(i), f[i].ctypes.data_as(C.POINTER(C.c_float)), lists[i].ctypes.data_as(C.POINTER(C.c_int32)),
                     lens[i].ctypes.data_as(C.POINTER(C.c_int32)), pid[i:i + 1].ctypes.data_as(C.POINTER(C.c_int32)))
    return f, lists, lens, pid


def test_obs_parity_vs_oracle(oracle, hip_lib):
    n, seed = 320, 4
    env = _env(n, seed)
    ob = oracle.OracleBatch(n, seed)
    done = 0
    for chunk in (0, 3, 40, 400, 1200):
        if chunk:
            env.random_rollout(done, chunk); ob.run_random(chunk, want_blobs=False); done += chunk
        f, lists, lens = env.get_obs()
        of, olists, olens, opid = _oracle_obs(ob, n)
        f = f.cpu().numpy()
        assert np.array_equal(f, of), f"obs floats differ after {done} steps: game {np.flatnonzero((f != of).any(1))[:4]} idx {np.flatnonzero((f != of).any(0))[:8]}"
        assert np.array_equal(lists.cpu().numpy(), olists) and np.array_equal(lens.cpu().numpy(), olens)
        assert np.array_equal(env.deciding_player().cpu().numpy(), opid)


def test_obs_vs_reference_states(hip_lib):
    """Reference observations at the sampled steps of the golden trajectories (states imported into the device)."""
    blobs, obs, lists, lens = [], [], [], []
    for name in gu.TRAJS:
        t = gu.load(name)
        o = gu.decode_obs(t)
        for k in range(len(t["sample_idx"])):
            blobs.append(t["sample_blob"][k].astype(np.int32)); obs.append(o[k])
            lists.append(t["sample_lists"][k].astype(np.int32)); lens.append(t["sample_lens"][k].astype(np.int32))
    env = _env(len(blobs), 0)
    env.import_state(np.array(blobs))
    f, gl, gn = env.get_obs()
    assert np.array_equal(f.cpu().numpy(), np.array(obs))
    assert np.array_equal(gl.cpu().numpy(), np.array(lists)) and np.array_equal(gn.cpu().numpy(), np.array(lens))


def test_obs_rows_bf16_and_storage_rows(oracle, hip_lib):
    """catan_obs_rows: (1) the bf16 dense matrix equals the fp32 one exactly (every observation value is a multiple of 1/8 below
    32), at game counts that are not multiples of the 16 games a wave takes; (2) the rows appended to a rollout storage -
    obs_f[t[g]][g], int8 lists / lengths - are the dense rows of the selected games, at every alignment a row can start at, and
    nothing else in the storage is touched; (3) the oracle's observations once more through this entry point."""
    import torch
    for n, seed, steps in ((333, 6, 700), (1024, 2, 1500), (5, 1, 40)):
        env = _env(n, seed)
        ob = oracle.OracleBatch(n, seed)
        env.random_rollout(0, steps); ob.run_random(steps, want_blobs=False)
        of, olists, olens, _ = _oracle_obs(ob, n)
        f32, l32, n32 = env.get_obs_rows(torch.float32)
        assert np.array_equal(f32.cpu().numpy(), of) and np.array_equal(l32.cpu().numpy(), olists) and np.array_equal(n32.cpu().numpy(), olens)
        fb, lb, nb = env.get_obs_rows(torch.bfloat16)
        assert torch.equal(fb.float(), f32) and torch.equal(lb, l32) and torch.equal(nb, n32)
        assert float(f32.max()) < 32.0 and torch.equal((f32 * 8).round(), f32 * 8)
        for dtype, dense in ((torch.bfloat16, fb), (torch.float32, f32)):
            S = 7
            g = torch.Generator().manual_seed(n)
            t = torch.randint(0, S, (n,), generator=g).cuda()
            sel = (torch.rand(n, generator=g) < 0.4).cuda()
            rows_f = torch.full((S, n, spec.OBS_FLOATS), -3.0, dtype=dtype, device="cuda")
            rows_l = torch.full((S, n, 5, 25), -7, dtype=torch.int8, device="cuda")
            rows_n = torch.full((S, n, 5), -7, dtype=torch.int8, device="cuda")
            out = env.get_obs_rows(dtype, rows=(rows_f, rows_l, rows_n), t=t, sel=sel)
            assert torch.equal(out[0], dense)
            ar = torch.arange(n, device="cuda")
            touched = torch.zeros((S, n), dtype=torch.bool, device="cuda")
            touched[t[sel], ar[sel]] = True
            assert torch.equal(rows_f[t[sel], ar[sel]], dense[sel]), "appended observation rows differ from the dense rows"
            assert torch.equal(rows_l[t[sel], ar[sel]].int(), l32[sel]) and torch.equal(rows_n[t[sel], ar[sel]].int(), n32[sel])
            assert bool((rows_f[~touched] == -3.0).all()) and bool((rows_l[~touched] == -7).all()) and bool((rows_n[~touched] == -7).all())
            # rows only (no dense output)
            rows_f2 = torch.full_like(rows_f, -3.0); rows_l2 = torch.full_like(rows_l, -7); rows_n2 = torch.full_like(rows_n, -7)
            assert env.get_obs_rows(dtype, rows=(rows_f2, rows_l2, rows_n2), t=t, sel=sel, dense=False) is None
            assert torch.equal(rows_f2, rows_f) and torch.equal(rows_l2, rows_l) and torch.equal(rows_n2, rows_n)


def test_env_wrapper_shim_signatures(oracle, hip_lib):
    """Single-game view with the reference's EnvWrapper signatures (env/wrapper.py:30-50,168-185,711-721)."""
    from settlers_of_catan_rl_amd.env import EnvWrapper
    env = EnvWrapper(seed=6, env_id=2)
    o = oracle.OracleEnv(6, 2); o.reset()
    obs = env.reset()
    assert set(spec.OBS_KEYS) <= set(obs) and obs["tile_representations"][0].shape == (60,)
    for step in range(300):
        masks = env.get_action_masks()
        assert len(masks) == 12 and [m.shape for m in masks] == list(spec.MASK_SHAPES)
        assert np.array_equal(np.concatenate([m.reshape(-1) for m in masks]).astype(np.float32), o.masks())
        a = o.sample_action(6, 2, step)
        heads = [a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7:11], a[11:15], a[15], a[16], a[17]]
        obs, rew, done, info = env.step(heads)
        orew, odone = o.step(a)
        assert done == odone and [rew[p] for p in (1, 2, 3, 4)] == list(orew)
        assert env.game.players_go == o.export()[spec.STATE_OFFSETS["players_go"][0]]
        if done:
            break
    st = env.save_state()
    m0 = env.get_action_masks()[0]
    illegal_type = int(np.flatnonzero(m0 == 0)[0])
    with pytest.raises(RuntimeError):                      # reference env/wrapper.py:38-41
        env.step([illegal_type] + [0] * 6 + [[0] * 4, [0] * 4, 0, 0, 0])
    assert np.array_equal(env.save_state()["blob"], st["blob"])          # a rejected action leaves the state untouched
    legal_type = int(np.flatnonzero(m0 > 0)[0])
    a = o.sample_action(6, 2, 10_000)
    env.step([a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7:11], a[11:15], a[15], a[16], a[17]])
    assert not np.array_equal(env.save_state()["blob"], st["blob"])
    env.restore_state(st)
    assert np.array_equal(env.save_state()["blob"], st["blob"]) and legal_type >= 0
    # the remaining attributes the reference's callers read: evaluation + forward search
    ob = o.export()
    assert env.curr_vps == {p: int(spec.state_field(ob, "curr_vps")[p - 1]) for p in (1, 2, 3, 4)}
    assert env.game.initial_settlements_placed == {p: int(spec.state_field(ob, "init_settlements")[p - 1]) for p in (1, 2, 3, 4)}
    assert env.game.initial_roads_placed == {p: int(spec.state_field(ob, "init_roads")[p - 1]) for p in (1, 2, 3, 4)}
    assert (env.winner is None) == (int(spec.state_field(ob, "winner")[0]) == 0)
    ctrl = env.game.players_go
    env.game.randomise_uncertainty(ctrl)                     # forward_search_policy/worker.py:46
    o.randomise_uncertainty(ctrl)
    assert np.array_equal(env.save_state()["blob"], o.export())


def test_longest_path_cases_vs_reference(hip_lib):
    g = gu.load("longest_road.npz")
    n = len(g["length"])
    blobs = np.zeros((n, spec.STATE_WORDS), dtype=np.int32)
    po = spec.STATE_OFFSETS
    for i in range(n):
        b = blobs[i]
        b[po["edge_owner"][0]:po["edge_owner"][0] + 72] = g["edge_owner"][i]
        co = g["corner_owner"][i].astype(np.int32)
        b[po["corner_owner"][0]:po["corner_owner"][0] + 54] = co
        b[po["corner_bld"][0]:po["corner_bld"][0] + 54] = (co > 0).astype(np.int32)
        b[po["player_order"][0]:po["player_order"][0] + 4] = [1, 2, 3, 4]
        b[po["players_go"][0]] = 1
        b[po["init_second_corner"][0]:po["init_second_corner"][0] + 4] = -1
    env = _env(n, 0)
    env.import_state(blobs)
    got = env.longest_path(g["player"].astype(np.int32)).cpu().numpy()
    assert np.array_equal(got, g["length"].astype(np.int32))


def test_longest_road_heavy_tier_matches_oracle(oracle, hip_lib):
    """Dense road networks that overflow the in-step (tier-1) budget and go through k_lr_heavy: the road-placement
    step must still produce the oracle's state."""
    import torch
    rng = np.random.default_rng(3)
    n = 64
    base = oracle.OracleBatch(n, 77)
    base.run_random(40)                                  # past initial placement for most games
    blobs = base.export()
    po = spec.STATE_OFFSETS
    keep = []
    for i in range(n):
        b = blobs[i]
        if b[po["initial_phase"][0]] or b[po["need_discard"][0]] or b[po["must_respond"][0]] or b[po["road_building_active"][0]] or b[po["just_moved_robber"][0]]:
            continue
        pid = int(b[po["players_go"][0]])
        eo = b[po["edge_owner"][0]:po["edge_owner"][0] + 72]
        free = np.flatnonzero(eo == 0)
        take = rng.choice(free, size=min(len(free), 44), replace=False)     # a dense 40+ edge network for the mover
        eo[take] = pid
        b[po["dice_rolled"][0]] = 1
        res = b[po[f"p{pid}_res"][0]:po[f"p{pid}_res"][0] + 5]
        bank = b[po["bank_res"][0]:po["bank_res"][0] + 5]
        for r in (0, 1):                                                     # make sure a road is affordable
            if res[r] == 0 and bank[r] > 0:
                res[r] += 1; bank[r] -= 1
        keep.append(i)
    assert len(keep) >= 16
    blobs = blobs[keep]
    env = _env(len(keep), 0, auto_reset=False)
    env.import_state(blobs)
    masks = env.get_action_masks().cpu().numpy()
    acts = np.zeros((len(keep), spec.ACTION_WORDS), dtype=np.int32)
    orcs = []
    for j in range(len(keep)):
        o = oracle.OracleEnv(0, j); o.import_(blobs[j]); orcs.append(o)
        assert np.array_equal(o.masks(), masks[j])
        road = np.flatnonzero(masks[j][spec.MASK_OFFSETS[2]:spec.MASK_OFFSETS[2] + 72])
        if masks[j][1] > 0 and len(road):
            acts[j, 0] = 1; acts[j, 2] = road[0]
        else:
            acts[j, 0] = 10                                                   # EndTurn
    rew, done = env.step(torch.from_numpy(acts))
    got = env.export_state().cpu().numpy()
    nroad = 0
    for j, o in enumerate(orcs):
        orew, odone = o.step(acts[j])
        nroad += int(acts[j, 0] == 1)
        assert np.array_equal(got[j], o.export()), spec.describe_state_diff(o.export(), got[j])
        assert np.array_equal(rew[j].cpu().numpy(), orew) and bool(done[j].item()) == odone
    assert nroad >= 8 and env.invalid_action_count() == 0


def test_gae_vs_reference_and_oracle(oracle, hip_lib):
    import torch
    from settlers_of_catan_rl_amd import ppo
    g = gu.load("gae_ppo.npz")
    for ci in range(3):
        r, v, m = (torch.from_numpy(g[f"gae{ci}_{k}"]).cuda() for k in ("rewards", "values", "masks"))
        ret, adv = ppo.compute_gae(r, v, m, 0.999, 0.95, process_group=False)
        oret, oadv = oracle.gae(g[f"gae{ci}_rewards"], g[f"gae{ci}_values"], g[f"gae{ci}_masks"], 0.999, 0.95)
        gr = ret.cpu().numpy()
        exact_ref = np.array_equal(gr, g[f"gae{ci}_returns"])
        exact_orc = np.array_equal(gr, oret)
        print(f"gae case {ci}: bit-identical to torch reference: {exact_ref}; to oracle: {exact_orc}; "
              f"max rel diff vs reference {np.max(np.abs(gr - g[f'gae{ci}_returns']) / (1 + np.abs(gr))):.2e}")
        assert np.allclose(gr, g[f"gae{ci}_returns"], rtol=2e-6, atol=1e-4)   # fp32 recurrence, same operand order
        assert np.allclose(gr, oret, rtol=2e-6, atol=1e-4)
        assert np.allclose(adv.cpu().numpy(), g[f"gae{ci}_adv"], rtol=1e-4, atol=1e-5)
        assert np.allclose(adv.cpu().numpy(), oadv, rtol=1e-4, atol=1e-5)
    # full-size property (cfg 3 shape): returns - adv_raw == values[:-1]; normalised advantages have mean 0 / std 1
    T, N = 200, 65536
    gen = torch.Generator(device="cuda").manual_seed(1)
    r = (torch.rand(T, N, device="cuda", generator=gen) < 0.01).float() * 500
    m = (torch.rand(T + 1, N, device="cuda", generator=gen) > 0.01).float()
    v = 150 + 50 * torch.randn(T + 1, N, device="cuda", generator=gen)
    ret, adv = ppo.compute_gae(r, v, m, 0.999, 0.95, process_group=False)
    assert abs(float(adv.double().mean())) < 1e-4 and abs(float(adv.double().std()) - 1.0) < 1e-3
    ret2, raw = ppo.compute_gae(r, v, m, 0.999, 0.95, process_group=False, normalise=False)
    assert torch.equal(ret2 - v[:-1], raw)


def test_ppo_loss_vs_reference(oracle, hip_lib):
    import torch
    from settlers_of_catan_rl_amd import ppo
    g = gu.load("gae_ppo.npz")
    for ci in range(2):
        t = {k: torch.from_numpy(g[f"ppo{ci}_{k}"]).cuda() for k in ("logp", "old", "adv", "v", "v_old", "ret")}
        logp = t["logp"].clone().requires_grad_(True); v = t["v"].clone().requires_grad_(True)
        total, losses = ppo.ppo_loss(logp, v, t["old"], t["adv"], t["v_old"], t["ret"], 0.2, 1.0)
        total.backward()
        assert abs(float(losses[0]) - float(g[f"ppo{ci}_action_loss"])) < 1e-5       # north_star tolerance
        assert abs(float(losses[1]) - float(g[f"ppo{ci}_value_loss"])) < 1e-5
        assert np.allclose(logp.grad.cpu().numpy(), g[f"ppo{ci}_dlogp"], atol=1e-6)
        assert np.allclose(v.grad.cpu().numpy(), g[f"ppo{ci}_dv"], atol=1e-6)
    # the reference's OWN PPO.update (ppo.py:26-79, driven by tools/gen_golden.py with stub storage / actor-critic): losses as it
    # returns them and the gradients its backward hands to the values / log-probs, value normaliser (lines 46-48) on and off
    for ci in range(3):
        t = {k: torch.from_numpy(g[f"ppoU{ci}_{k}"]).cuda() for k in ("logp", "old", "adv", "v", "v_old", "ret")}
        logp = t["logp"].clone().requires_grad_(True); v = t["v"].clone().requires_grad_(True)
        coef = float(g["ppoU_value_loss_coef"])
        total, losses = ppo.ppo_loss(logp, v, t["old"], t["adv"], t["v_old"], t["ret"], float(g["ppoU_clip"]), coef,
                                     value_normaliser=(150.0, 150.0) if int(g[f"ppoU{ci}_use_norm"]) else None)
        total.backward()
        assert abs(float(losses[0]) - float(g[f"ppoU{ci}_action_loss"])) < 1e-5
        assert abs(float(losses[1]) * coef - float(g[f"ppoU{ci}_value_loss_x_coef"])) < 1e-5
        assert np.allclose(logp.grad.cpu().numpy(), g[f"ppoU{ci}_dlogp"], atol=1e-7, rtol=1e-4)
        assert np.allclose(v.grad.cpu().numpy(), g[f"ppoU{ci}_dv"], atol=1e-7, rtol=1e-3)
    # value normaliser path (ppo.py:46-48) against plain torch fp32 at the cfg-3 minibatch size
    B = 204800
    gen = torch.Generator(device="cuda").manual_seed(2)
    logp = (torch.randn(B, 1, device="cuda", generator=gen) * 0.4 - 3).requires_grad_(True)
    old = logp.detach() + 0.3 * torch.randn(B, 1, device="cuda", generator=gen)
    adv = torch.randn(B, 1, device="cuda", generator=gen)
    v = torch.randn(B, 1, device="cuda", generator=gen).requires_grad_(True)
    vp = 150 + 150 * (v.detach() + 0.3 * torch.randn(B, 1, device="cuda", generator=gen))
    ret = 150 + 150 * torch.randn(B, 1, device="cuda", generator=gen)
    total, losses = ppo.ppo_loss(logp, v, old, adv, vp, ret, 0.2, 1.0, value_normaliser=(150.0, 150.0))
    total.backward()
    gl, gv = logp.grad.clone(), v.grad.clone()
    logp.grad = None; v.grad = None
    vpn, retn = (vp - 150.0) / (150.0 + 1e-4), (ret - 150.0) / (150.0 + 1e-4)
    ratio = torch.exp(logp - old)
    al = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
    vpc = vpn + (v - vpn).clamp(-0.2, 0.2)
    vl = 0.5 * torch.max((v - retn).pow(2), (vpc - retn).pow(2)).mean()
    (vl + al).backward()
    assert abs(float(losses[0]) - float(al)) < 1e-5 and abs(float(losses[1]) - float(vl)) < 1e-5
    assert torch.allclose(gl, logp.grad, atol=1e-8, rtol=1e-4) and torch.allclose(gv, v.grad, atol=1e-8, rtol=1e-4)
