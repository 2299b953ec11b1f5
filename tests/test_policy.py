"""Policy / value net: parity with the upstream reference net under identical weights (runs where /root/reference is
mounted, i.e. the development container), plus reference-free consistency checks."""
import os
import sys

import numpy as np
import pytest
import torch

from settlers_of_catan_rl_amd import spec
from settlers_of_catan_rl_amd.policy import CatanPolicy
import policy_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/RL/models")


def _perturb(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)


def test_parameter_inventory():
    m = CatanPolicy()
    assert sum(p.numel() for p in m.parameters()) == 1928995 - 2      # reference total minus value_normaliser.mean/std
    keys = set(m.state_dict())
    for k in ("observation_module.tile_encoder.encoder_layers.1.multi_headed_attention.qkv_nets.2.weight",
              "observation_module.other_players_module.final_linear_layer.weight",
              "action_head_module.action_heads.5.custom_mlp.weight", "action_head_module.action_heads.8.mlp_1.weight",
              "action_head_module.action_heads.10.distribution.linear.bias", "value_out.weight", "v_norm_2.bias"):
        assert k in keys
    assert m.state_dict()["action_head_module.action_heads.10.mlp_1.weight"].shape == (128, 521)


def test_act_evaluate_consistency(oracle):
    torch.manual_seed(0)
    m = CatanPolicy(); _perturb(m)
    x = policy_util.oracle_batch_inputs(oracle, n=32)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        v, a, lp = m.act(x["obs_f"], x["lists"], x["lens"], x["masks"], generator=g)
        v2, lp2, ent = m.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], a)
    assert torch.allclose(v, v2) and torch.allclose(lp, lp2, atol=1e-6) and torch.isfinite(lp).all() and float(ent) > 0
    # every sampled head value is legal under the env masks that apply to the sampled type
    masks = x["masks"].numpy(); a = a.numpy()
    for i in range(len(a)):
        t = a[i, 0]
        assert masks[i, t] == 1
        if t == 0: assert masks[i, spec.MASK_OFFSETS[1] + a[i, 1]] == 1
        if t == 2: assert masks[i, spec.MASK_OFFSETS[1] + 54 + a[i, 1]] == 1
        if t == 1: assert masks[i, spec.MASK_OFFSETS[2] + a[i, 2]] == 1
        if t == 8: assert masks[i, spec.MASK_OFFSETS[3] + a[i, 3]] == 1
        if t == 12: assert masks[i, spec.MASK_OFFSETS[11] + a[i, 17]] == 1
        if t == 6: assert a[i, 7] > 0 and x["obs_f"][i, 12 + a[i, 7]] > 0        # first give resource is owned
    # gradients reach every parameter through value + log-prob + entropy
    v, lp, ent = m.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], torch.from_numpy(a))
    (v.mean() + lp.mean() + ent).backward()
    dead = [k for k, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not dead, dead


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_parity_with_reference_net(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ref_bootstrap import bootstrap
    bootstrap()
    from RL.models.build_agent_model import build_agent_model
    torch.manual_seed(1)
    ref = build_agent_model()
    _perturb(ref, seed=2)
    ref.eval()
    mine = CatanPolicy()
    mine.load_reference_state_dict(ref.state_dict())
    mine.eval()
    x = policy_util.oracle_batch_inputs(oracle, n=40, seed=8)
    B = x["obs_f"].shape[0]
    o = spec.OBS_FLOAT_OFFSETS
    obs = {k: x["obs_f"][:, o[k]:o[k] + int(np.prod(shp))].reshape((B,) + shp).clone() for k, shp in spec.OBS_FLOAT_KEYS.items()}
    for i, k in enumerate(spec.OBS_LIST_KEYS):
        obs[k] = x["lists"][:, i].long()
    masks = []
    for hi, (off, sz, shp) in enumerate(zip(spec.MASK_OFFSETS, spec.MASK_SIZES, spec.MASK_SHAPES)):
        mk = x["masks"][:, off:off + sz].reshape((B,) + shp).clone()
        masks.append(mk.transpose(0, 1).contiguous() if hi in (1, 6, 9) else mk)
    with torch.no_grad():
        # deterministic act: identical arg-max actions, values and log-probs
        v_m, a_m, lp_m = mine.act(x["obs_f"], x["lists"], x["lens"], x["masks"], deterministic=True)
        v_r, a_r, lp_r, _ = ref.act({k: v.clone() for k, v in obs.items()}, None, None, [mk.clone() for mk in masks], deterministic=True)
        a_r_flat = torch.cat([torch.stack([t.view(-1) for t in h], 1) if isinstance(h, list) else h.view(B, -1) for h in a_r], 1)
        assert torch.allclose(v_m, v_r, atol=1e-5), float((v_m - v_r).abs().max())
        assert torch.equal(a_m, a_r_flat)
        assert torch.allclose(lp_m, lp_r, atol=1e-5), float((lp_m - lp_r).abs().max())
        # evaluate_actions on sampled (non-greedy) actions incl. the quirks of the recurrent trade heads
        g = torch.Generator().manual_seed(11)
        _, a_s, _ = mine.act(x["obs_f"], x["lists"], x["lens"], x["masks"], generator=g)
        acts_ref = [a_s[:, off:off + ln].clone() for off, ln in spec.ACTION_HEAD_SLICES]
        v_r, lp_r, ent_r, _ = ref.evaluate_actions({k: v.clone() for k, v in obs.items()}, None, None, acts_ref, [mk.clone() for mk in masks])
        v_m, lp_m, ent_m = mine.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], a_s)
        assert torch.allclose(v_m, v_r, atol=1e-5)
        assert torch.allclose(lp_m, lp_r, atol=1e-5), float((lp_m - lp_r).abs().max())
        assert abs(float(ent_m) - float(ent_r)) < 1e-5


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_lstm_path_parity_with_reference_net(oracle):
    """include_lstm (build_agent_model.py:26 switched on): one step per row, and the truncated-BPTT form (T steps of B
    sequences with terminal masks inside), against the reference net with identical weights."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ref_bootstrap import bootstrap
    bootstrap()
    import RL.models.build_agent_model as bam
    old = bam.include_lstm
    bam.include_lstm = True
    try:
        torch.manual_seed(3)
        ref = bam.build_agent_model()
    finally:
        bam.include_lstm = old
    _perturb(ref, seed=4)
    ref.eval()
    mine = CatanPolicy(include_lstm=True)
    mine.load_reference_state_dict(ref.state_dict())
    mine.eval()
    assert sum(p.numel() for p in mine.parameters()) == sum(p.numel() for p in ref.parameters()) - 2
    T, Bs = 5, 8
    x = {k: v[:T * Bs] for k, v in policy_util.oracle_batch_inputs(oracle, n=T * Bs, seed=9).items()}
    B = x["obs_f"].shape[0]
    assert B == T * Bs
    o = spec.OBS_FLOAT_OFFSETS
    obs = {k: x["obs_f"][:, o[k]:o[k] + int(np.prod(shp))].reshape((B,) + shp).clone() for k, shp in spec.OBS_FLOAT_KEYS.items()}
    for i, k in enumerate(spec.OBS_LIST_KEYS):
        obs[k] = x["lists"][:, i].long()
    masks = []
    for hi, (off, sz, shp) in enumerate(zip(spec.MASK_OFFSETS, spec.MASK_SIZES, spec.MASK_SHAPES)):
        mk = x["masks"][:, off:off + sz].reshape((B,) + shp).clone()
        masks.append(mk.transpose(0, 1).contiguous() if hi in (1, 6, 9) else mk)
    g = torch.Generator().manual_seed(21)
    cp = lambda d: {k: v.clone() for k, v in d.items()}
    with torch.no_grad():
        # (1) one step per row, random incoming state, some rows with a zero terminal mask
        h0 = torch.randn(B, 256, generator=g) * 0.5; c0 = torch.randn(B, 256, generator=g) * 0.5
        nt = (torch.rand(B, 1, generator=g) > 0.3).float()
        v_m, a_m, lp_m, (h_m, c_m) = mine.act(x["obs_f"], x["lists"], x["lens"], x["masks"], deterministic=True, hidden=(h0, c0), nonterminal=nt)
        v_r, a_r, lp_r, (h_r, c_r) = ref.act(cp(obs), (h0.clone(), c0.clone()), nt.clone(), [mk.clone() for mk in masks], deterministic=True)
        a_r_flat = torch.cat([torch.stack([t.view(-1) for t in h], 1) if isinstance(h, list) else h.view(B, -1) for h in a_r], 1)
        assert torch.allclose(h_m, h_r, atol=1e-5) and torch.allclose(c_m, c_r, atol=1e-5)
        assert torch.allclose(v_m, v_r, atol=1e-5), float((v_m - v_r).abs().max())
        assert torch.equal(a_m, a_r_flat)
        assert torch.allclose(lp_m, lp_r, atol=1e-5)
        assert torch.allclose(mine.get_value(x["obs_f"], x["lists"], x["lens"], (h0, c0), nt), ref.get_value(cp(obs), (h0.clone(), c0.clone()), nt.clone()), atol=1e-5)
        # (2) T steps of Bs sequences, zeros inside the mask (incl. at t = 0 and two in the same step)
        hs = torch.randn(Bs, 256, generator=g) * 0.5; cs = torch.randn(Bs, 256, generator=g) * 0.5
        nts = torch.ones(T, Bs); nts[0, 1] = 0; nts[2, 3] = 0; nts[2, 5] = 0; nts[4, 0] = 0
        nts = nts.reshape(T * Bs, 1)
        _, a_s, _, _ = mine.act(x["obs_f"], x["lists"], x["lens"], x["masks"], generator=g, hidden=(h0, c0), nonterminal=nt)
        acts_ref = [a_s[:, off:off + ln].clone() for off, ln in spec.ACTION_HEAD_SLICES]
        v_r, lp_r, ent_r, (h_r, c_r) = ref.evaluate_actions(cp(obs), (hs.clone(), cs.clone()), nts.clone(), acts_ref, [mk.clone() for mk in masks])
        v_m, lp_m, ent_m, (h_m, c_m) = mine.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], a_s, hidden=(hs, cs), nonterminal=nts)
        assert torch.allclose(v_m, v_r, atol=2e-5), float((v_m - v_r).abs().max())
        assert torch.allclose(lp_m, lp_r, atol=2e-5), float((lp_m - lp_r).abs().max())
        assert abs(float(ent_m) - float(ent_r)) < 1e-5
        assert torch.allclose(h_m, h_r, atol=1e-5) and torch.allclose(c_m, c_r, atol=1e-5)
        # all-ones masks (the reference's scalar `has_zeros` branch)
        ones = torch.ones(T * Bs, 1)
        v_r, _, _, _ = ref.evaluate_actions(cp(obs), (hs.clone(), cs.clone()), ones.clone(), acts_ref, [mk.clone() for mk in masks])
        v_m, _, _, _ = mine.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], a_s, hidden=(hs, cs), nonterminal=ones)
        assert torch.allclose(v_m, v_r, atol=2e-5)
    # gradients flow through time into the LSTM weights
    v, lp, ent, _ = mine.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], a_s, hidden=(hs, cs), nonterminal=nts)
    (v.mean() + lp.mean() + ent).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0 for p in mine.lstm.parameters())


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_reference_state_dict_loads_strictly_into_the_reference_net(tmp_path):
    """The checkpoint tuple written by train_loop (central state-dict and league entries) must pass the reference's STRICT
    `load_state_dict` (robust_train.py:55-59, game_manager.py:161-162)."""
    from settlers_of_catan_rl_amd import train_loop as tl
    from settlers_of_catan_rl_amd.league import League
    from RL.models.build_agent_model import build_agent_model
    net = CatanPolicy()
    ref = build_agent_model()
    ref.load_state_dict(net.reference_state_dict(), strict=True)
    assert set(net.reference_state_dict()) == set(ref.state_dict())

    class _Env(object):
        n = 4
        def set_reward_annealing_factor(self, f): pass

    class _Col(object):
        N = 4
        def set_opponents(self, nets, idx): pass

    class _Tr(object):
        optimiser = torch.optim.Adam(net.parameters(), lr=3e-4)
        class cfg: entropy_coef = 0.0

    lg = League(envs_per_worker=2, seed=0)
    loop = tl.TrainingLoop(_Env(), net, _Col(), _Tr(), tl.TrainArgs(num_steps=2, total_env_steps=800), league=lg, make_net=CatanPolicy)
    lg.add(net)
    loop.save_reference_tuple(str(tmp_path / "ref.pt"))
    sd, earlier, _, _, _ = torch.load(str(tmp_path / "ref.pt"), weights_only=False)
    ref.load_state_dict(sd, strict=True)
    assert len(earlier) == 2
    for e in earlier:
        build_agent_model().load_state_dict(e, strict=True)
    # and the other way round
    loop.load_reference_tuple(str(tmp_path / "ref.pt"))


def test_compact_head_evaluation_equals_the_dense_one(oracle):
    """evaluate_actions runs every action head only on the rows whose action type uses it (`_evaluate_compact`): same joint
    log-probs, entropy and gradients as the dense evaluation of all heads on all rows."""
    torch.manual_seed(4)
    net = CatanPolicy()
    _perturb(net, seed=6)
    x = policy_util.oracle_batch_inputs(oracle, n=160, seed=21, steps=(0, 9, 60, 200, 500, 900, 1300, 1700))
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        _, acts, _ = net.act(x["obs_f"], x["lists"], x["lens"], x["masks"], generator=g)
    types = set(acts[:, 0].tolist())
    assert len(types) >= 11, types                       # nearly every action type occurs (incl. trades, cards, robber, discard)
    assert ((acts[:, 0] == 4) & (acts[:, 4] == 2)).any() or ((acts[:, 0] == 4) & (acts[:, 4] == 4)).any() or (acts[:, 0] == 5).any()
    res = {}
    net.action_head_module.compact_min_rows = 0
    for compact in (False, True):
        net.action_head_module.compact_evaluate = compact
        net.zero_grad()
        v, lp, ent = net.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], acts)
        ((lp[:, 0] * torch.linspace(0.5, 1.5, lp.shape[0])).sum() + 3.0 * ent + v.sum()).backward()
        res[compact] = (lp.detach().clone(), float(ent), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    net.action_head_module.compact_evaluate = True
    assert torch.allclose(res[True][0], res[False][0], atol=1e-5)
    assert abs(res[True][1] - res[False][1]) < 1e-6
    assert set(res[True][2]) == set(res[False][2])
    for k, gd in res[False][2].items():
        assert torch.allclose(res[True][2][k], gd, atol=2e-5, rtol=1e-4), (k, float((res[True][2][k] - gd).abs().max()))


@pytest.mark.parametrize("which", ["ff", "lstm"])
def test_policy_fixture_from_the_reference_net(which):
    """tests/golden/policy_small.npz (tools/gen_golden.py gen_policy_small: the reference's own net on real observations):
    identical arg-max actions; value / joint log-prob / entropy / LSTM state within 1e-5; every parameter's gradient (norm and
    a hashed projection) within 1e-4 of its size.  The same fixture is checked with the HIP kernels on under -m gpu."""
    import golden_util as gu
    import policy_fixture as pf
    dev = pf.check_policy_fixture(gu.load("policy_small.npz"), which, "cpu", tol=1e-5, grad_tol=1e-4)
    assert dev["act_argmax_agreement"] == 1.0
