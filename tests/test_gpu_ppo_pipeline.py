"""GPU: the learner-side pipeline end to end (config 3 shape, tiny sizes): device rollout collection -> values -> GAE
kernel -> minibatches -> fused PPO loss -> Adam.  Checks against plain-torch fp32 restatements of the reference formulas."""
import numpy as np
import pytest
import torch

from settlers_of_catan_rl_amd import spec

pytestmark = pytest.mark.gpu


def test_rollout_and_update_on_device(hip_lib):
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    from settlers_of_catan_rl_amd import ppo as K
    torch.manual_seed(0)
    N, T = 256, 12
    env = VecCatanEnv(N, seed=5)
    env.random_rollout(0, 1500)                     # mid/late games so that some finish inside the rollout
    net = CatanPolicy().cuda()
    col = RolloutCollector(env, net, T, seed=1, autocast_dtype=None)
    st = col.gather_rollouts()
    assert env.invalid_action_count() == 0, "the policy must only emit mask-legal actions"
    assert int(col.n_obs.min()) == T + 1 and int(col.n_act.min()) >= T
    assert torch.isfinite(st.action_log_probs).all() and (st.action_log_probs <= 0).all()
    # stored action masks really are the masks of the stored observations' states: the chosen type is legal in them
    am = st.unpack_action_masks(st.action_masks)
    typ = st.actions[..., 0]
    assert (am.gather(-1, typ[..., None]) == 1).all()
    # ---- one epoch by hand in fp32 vs the trainer's kernels
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=4), autocast_dtype=None, seed=3)
    values = tr.compute_values(st)
    ret, adv = K.compute_gae(st.rewards[:T].contiguous(), values, st.masks[:T + 1].contiguous(), 0.999, 0.95, process_group=False)
    gae = torch.zeros(N, device="cuda"); ret_ref = torch.zeros_like(ret)
    for t in reversed(range(T)):                     # RL/ppo/process_batch.py:134-139
        delta = st.rewards[t] + 0.999 * values[t + 1] * st.masks[t + 1] - values[t]
        gae = delta + 0.999 * 0.95 * st.masks[t + 1] * gae
        ret_ref[t] = gae + values[t]
    a_ref = ret_ref - values[:-1]
    a_ref = (a_ref - a_ref.mean()) / (a_ref.std() + 1e-5)
    assert torch.allclose(ret, ret_ref, rtol=1e-5, atol=1e-3) and torch.allclose(adv, a_ref, rtol=1e-4, atol=1e-4)
    before = [p.detach().clone() for p in net.parameters()]
    vl, al, el = tr.update(st)
    assert all(np.isfinite(x) for x in (vl, al, el))
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(before, net.parameters()))
    assert changed > 100
    col.after_rollouts()
    st2 = col.gather_rollouts()                       # second rollout continues from the carried observation
    assert torch.isfinite(st2.obs_f.float()).all() and env.invalid_action_count() == 0


def test_lstm_policy_rollout_and_bptt_update_on_device(hip_lib):
    """`include_lstm` end to end on the device - per-seat LSTM states in the rollout, the stored state of
    every decision equals a replay of the state recurrence, one-step value re-evaluation, truncated-BPTT minibatches."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    N, T, L = 128, 20, 10
    env = VecCatanEnv(N, seed=6)
    env.random_rollout(0, 1500)
    net = CatanPolicy(include_lstm=True).cuda()
    col = RolloutCollector(env, net, T, seed=2, autocast_dtype=None)
    st = col.gather_rollouts()
    assert env.invalid_action_count() == 0 and st.hidden.shape == (2, T + 1, N, 256)
    assert float(st.hidden[:, 1:].abs().max()) > 0.01 and torch.isfinite(st.hidden).all()
    # (in the first rollout after reset() the reference's terminal masks are shifted by one against the observations for
    # every game whose active seat does not move first - game_manager.py:49 "IS THIS RIGHT??" - and aligned afterwards)
    col.after_rollouts()
    st = col.gather_rollouts()
    # the state stored for decision t+1 is the LSTM step of decision t from its stored state - unless the game ended in
    # between (mask 0 -> zero state)
    with torch.no_grad():
        f = st.obs_f[:T].reshape(T * N, -1).float(); lists = st.lists[:T].reshape(T * N, 5, -1); lens = st.lens[:T].reshape(T * N, 5).long()
        hid = (st.hidden[0, :T].reshape(T * N, -1), st.hidden[1, :T].reshape(T * N, -1))
        _, _, (h1, c1) = net.base(f, lists, lens, hid, st.masks[:T].reshape(T * N))
        nt = st.masks[1:T + 1].reshape(T * N, 1)
        assert torch.allclose(h1 * nt, st.hidden[0, 1:T + 1].reshape(T * N, -1), atol=1e-4)
        assert torch.allclose(c1 * nt, st.hidden[1, 1:T + 1].reshape(T * N, -1), atol=1e-4)
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=4, truncated_seq_len=L), autocast_dtype=torch.bfloat16, seed=3)
    v = tr.compute_values(st)
    assert v.shape == (T + 1, N) and torch.isfinite(v).all()
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    vl, al, el = tr.update(st)
    assert all(np.isfinite(x) for x in (vl, al, el))
    assert all(not torch.equal(before[k], p) for k, p in net.named_parameters() if k.startswith("lstm."))
    col.after_rollouts()
    st2 = col.gather_rollouts()
    assert torch.isfinite(st2.hidden).all() and env.invalid_action_count() == 0


def test_bf16_autocast_forward_close_to_fp32(hip_lib):
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    torch.manual_seed(0)
    env = VecCatanEnv(512, seed=2)
    env.random_rollout(0, 700)
    f, lists, lens = env.get_obs()
    masks = env.get_action_masks()
    net = CatanPolicy().cuda()
    with torch.no_grad():
        v32, a, lp32 = net.act(f, lists, lens.long(), masks, deterministic=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            v16, lp16, _ = net.evaluate_actions(f, lists, lens.long(), masks, a)
    assert float((v32 - v16).abs().max()) < 0.1 and float((lp32 - lp16).abs().max()) < 0.25


@pytest.mark.parametrize("L,H,HD", [(19, 4, 16), (25, 4, 4)])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fused_attention_kernel_vs_torch(hip_lib, L, H, HD, dtype):
    """csrc/catan_nn.hip against the reference formulation (multi_headed_attention.py:25-36) in fp32 torch ops."""
    import math
    from settlers_of_catan_rl_amd import nn_kernels
    dt = getattr(torch, dtype)
    torch.manual_seed(0)
    for B in (1, 5, 1001):
        qkv = torch.randn(B, L, 3, H, HD, device="cuda").to(dt).requires_grad_(True)
        lens = torch.randint(1, L + 1, (B,), device="cuda", dtype=torch.int32) if L == 25 else None
        out = nn_kernels.small_attention(qkv, lens)
        go = torch.randn_like(out.float()).to(dt)
        out.backward(go)
        g1 = qkv.grad.float().clone()
        x = qkv.detach().float().requires_grad_(True)
        q, k, v = x.permute(2, 0, 3, 1, 4)
        s = q @ k.transpose(-2, -1) / math.sqrt(HD)
        if lens is not None:
            km = torch.arange(L, device="cuda")[None, :] < lens[:, None]
            s = s.masked_fill(~km[:, None, None, :], float("-inf"))
        ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, H * HD)
        ref.backward(go.float())
        tol = 1e-5 if dtype == "float32" else 3e-2
        assert torch.allclose(out.float(), ref, atol=tol, rtol=tol), float((out.float() - ref).abs().max())
        assert torch.allclose(g1, x.grad, atol=tol * 4, rtol=tol * 4), float((g1 - x.grad).abs().max())


@pytest.mark.parametrize("D", [16, 25, 64, 128, 256, 512])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("relu", [False, True])
def test_small_layer_norm_kernel_vs_torch(hip_lib, D, dtype, relu):
    from settlers_of_catan_rl_amd import nn_kernels
    dt = getattr(torch, dtype)
    torch.manual_seed(1)
    ln = torch.nn.LayerNorm(D).cuda()
    with torch.no_grad():
        ln.weight.add_(0.3 * torch.randn(D, device="cuda")); ln.bias.add_(0.3 * torch.randn(D, device="cuda"))
    for shape in ((7, D), (333, 19, D), (70001, D)):
        x = (torch.randn(*shape, device="cuda") * 2 + 0.5).to(dt).requires_grad_(True)
        y = nn_kernels.small_layer_norm(x, ln, relu)
        go = torch.randn_like(y.float()).to(dt)
        ln.zero_grad()
        y.backward(go)
        gx, gw, gb = x.grad.float().clone(), ln.weight.grad.clone(), ln.bias.grad.clone()
        xr = x.detach().float().requires_grad_(True)
        ln.zero_grad()
        yr = ln(xr)
        yr = torch.relu(yr) if relu else yr
        yr.backward(go.float())
        tol = 2e-5 if dtype == "float32" else 4e-2
        assert torch.allclose(y.float(), yr, atol=tol, rtol=tol)
        bad = ~torch.isclose(gx, xr.grad, atol=tol * 4, rtol=tol * 4)
        # with the fused ReLU an output within rounding of 0 can fall on the other side of the threshold than in torch's
        # summation order, which changes that row's gradient: allow a vanishing fraction of such rows
        assert float(bad.float().mean()) <= (2e-5 if relu else 0.0)
        n = x.numel() // D
        assert torch.allclose(gw, ln.weight.grad, atol=tol * 4 * max(1, n ** 0.5), rtol=2e-2)
        assert torch.allclose(gb, ln.bias.grad, atol=tol * 4 * max(1, n ** 0.5), rtol=2e-2)


@pytest.mark.parametrize("R,I,O", [(100003, 60, 64), (5000, 6, 16), (70001, 64, 192), (33333, 152, 256), (4097, 16, 25),
                                   (20000, 128, 64), (64, 16, 48), (130, 159, 256), (3001, 8, 8), (100001, 64, 128), (50001, 128, 256), (777, 152, 8),
                                   (20001, 512, 128), (30001, 256, 256), (5000, 1024, 64), (777, 168, 8)])
def test_linear_wgrad_kernel_vs_torch(hip_lib, R, I, O):
    """k_wgrad / k_wgrad_tr (MFMA, rows split over the grid; widths that are multiples of 8 take the variant with the row-major
    LDS image and transposing LDS reads; inputs wider than 159 in column slices of 128): dw = dy^T x, db = column sums of dy, against fp32 torch on the same bf16 inputs.
    Asymmetric random data (a transposed or row/column-swapped result cannot pass)."""
    import ctypes as C
    from settlers_of_catan_rl_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(R + I)
    x = (torch.randn((R, I), device="cuda", generator=g) * torch.linspace(0.5, 2.0, I, device="cuda")).to(torch.bfloat16)
    dy = (torch.randn((R, O), device="cuda", generator=g) * torch.linspace(2.0, 0.25, O, device="cuda") + 0.1).to(torch.bfloat16)
    dw = torch.zeros((O, I), device="cuda")
    db = torch.zeros((O,), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.catan_linear_wgrad_supported(R, I, O)
    _lib.check(L.catan_linear_wgrad(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), C.c_void_p(db.data_ptr()),
                                    R, I, O, st))
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    tol = 2e-5 * float(ref_w.abs().max()) + 1e-3        # fp32 accumulation of exact bf16 products, different order
    assert float((dw.double() - ref_w).abs().max()) < tol
    assert float((db.double() - ref_b).abs().max()) < 2e-5 * float(ref_b.abs().max()) + 1e-3


def test_grouped_linear_wgrad_equals_the_single_launches(hip_lib):
    """catan_linear_wgrad_grouped: a mixed list of layers - sliced 512-wide inputs on ragged row segments (the heads' first layers),
    plain widths of three tile shapes, widths that are not multiples of 8 (launched on their own), with and without a bias
    gradient, more units than one launch holds - against fp32 torch and bit-for-bit against catan_linear_wgrad problem by problem
    (same blocks, same atomics order within a problem is not guaranteed: compared with a tolerance of a few ULPs of the sums)."""
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(1820, 512, 128), (6101, 512, 128), (37138, 512, 128), (40142, 512, 128), (13073, 512, 128), (4097, 512, 128), (16298, 512, 128),
              (36842, 512, 128), (12858, 512, 128), (3085, 512, 128), (4007, 512, 128), (20000, 128, 64), (70001, 64, 192), (33333, 152, 256),
              (5000, 6, 16), (100001, 64, 128), (777, 168, 8), (30001, 256, 256), (9000, 16, 48)]
    xs = [(torch.randn((R, I), device="cuda", generator=g) * torch.linspace(0.5, 2.0, I, device="cuda")).to(torch.bfloat16) for R, I, O in shapes]
    dys = [(torch.randn((R, O), device="cuda", generator=g) * torch.linspace(2.0, 0.25, O, device="cuda") + 0.1).to(torch.bfloat16) for R, I, O in shapes]
    got = nn_kernels.wgrad_grouped(list(zip(xs, dys)))
    for (R, I, O), x, dy, (dw, db) in zip(shapes, xs, dys, got):
        ref_w = dy.double().t() @ x.double()
        ref_b = dy.double().sum(0)
        assert float((dw.double() - ref_w).abs().max()) < 2e-5 * float(ref_w.abs().max()) + 1e-3, (R, I, O)
        assert float((db.double() - ref_b).abs().max()) < 2e-5 * float(ref_b.abs().max()) + 1e-3, (R, I, O)
        one_w, one_b = nn_kernels.wgrad(x, dy)
        assert float((dw - one_w).abs().max()) <= 1e-5 * float(ref_w.abs().max()) + 1e-4, (R, I, O)
    got_nb = nn_kernels.wgrad_grouped(list(zip(xs[:3], dys[:3])), has_bias=False)
    assert all(db is None for _, db in got_nb) and all(torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-3) for a, b in zip(got_nb, got[:3]))


def test_policy_grads_with_wgrad_kernel_match_library_path(hip_lib, monkeypatch):
    """The net's parameter gradients under bf16 autocast: tall-skinny Linear layers through k_wgrad vs through F.linear."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd import nn_kernels
    torch.manual_seed(1)
    B = 4096
    env = VecCatanEnv(B, seed=3)
    env.random_rollout(0, 400)
    f, lists, lens = env.get_obs()
    masks = env.get_action_masks()
    lens = lens.long()
    net = CatanPolicy().cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, a, _ = net.act(f, lists, lens, masks, generator=torch.Generator(device="cuda").manual_seed(0))

    def grads(use_kernel):
        if not use_kernel:
            monkeypatch.setattr(nn_kernels, "linear_supported", lambda x, w: False)
        net.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            v, lp, ent = net.evaluate_actions(f, lists, lens, masks, a)
        (v.float().mean() + lp.float().mean() - 0.01 * ent).backward()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    g1 = grads(True)
    g0 = grads(False)
    assert g0.keys() == g1.keys()
    floor = 1e-3 * max(float(g.norm()) for g in g0.values())     # (the key bias of an attention has a ZERO true gradient - the softmax
    for k in g0:                                                 #  ignores a constant added to every score: pure bf16 noise on both sides)
        den = float(g0[k].norm()) + floor
        assert float((g1[k] - g0[k]).norm()) / den < 5e-2, k     # bf16 activations; the kernel path keeps dw in fp32


def test_deferred_weight_gradients_equal_the_immediate_ones(hip_lib):
    """nn_kernels.wgrad_queue (PPOTrainer's steps): the tall-skinny layers' weight gradients queued during the backward pass and
    accumulated by grouped launches after it - into a `.grad` autograd created from the zero tensor it was handed (`.grad = None`
    before the backward: one rank), and into a pre-assigned buffer (the flat bucket's views under several ranks) - equal the
    gradients of the same backward with every weight gradient launched where it arises (fp32 sums in another order)."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd import nn_kernels
    torch.manual_seed(1)
    B = 16384
    env = VecCatanEnv(B, seed=3)
    env.random_rollout(0, 400)
    f, lists, lens = env.get_obs()
    masks = env.get_action_masks()
    lens = lens.long()
    net = CatanPolicy().cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, a, _ = net.act(f, lists, lens, masks, generator=torch.Generator(device="cuda").manual_seed(0))

    def grads(mode):
        for p in net.parameters():
            p.grad = torch.zeros_like(p) if mode == "preassigned" else None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            v, lp, ent = net.evaluate_actions(f, lists, lens, masks, a)
        if mode != "immediate":
            nn_kernels.wgrad_queue.begin()
        (v.float().mean() + lp.float().mean() - 0.01 * ent).backward()
        queued = len(nn_kernels.wgrad_queue.items)
        nn_kernels.wgrad_queue.flush()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}, queued

    g0, q0 = grads("immediate")
    assert q0 == 0
    for mode in ("deferred", "preassigned"):
        g1, q1 = grads(mode)
        assert q1 >= 20, q1                                          # (the layers really went through the queue)
        assert g0.keys() == g1.keys()
        floor = 1e-6 * max(float(g.norm()) for g in g0.values())
        for k in g0:
            assert float((g1[k] - g0[k]).norm()) <= 2e-5 * float(g0[k].norm()) + floor, (mode, k, float((g1[k] - g0[k]).norm()), float(g0[k].norm()))
    assert not nn_kernels.wgrad_queue.active and not nn_kernels.wgrad_queue.items


def test_rollout_with_league_opponents(hip_lib):
    """Per-worker league opponents (league.League.assign -> grouped inference in the collector): a rollout + update runs,
    no illegal action reaches the env, and with snapshots identical to the central policy the stored log-probs are those
    of the central policy on the stored observations."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.league import League
    torch.manual_seed(0)
    N, T = 200, 6
    env = VecCatanEnv(N, seed=9)
    env.random_rollout(0, 300)
    net = CatanPolicy().cuda()
    col = RolloutCollector(env, net, T, seed=1)
    lg = League(envs_per_worker=5, seed=2)
    lg.add(net)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))
    lg.add(net)                                              # two distinct snapshots in the deque
    distinct = lg.assign(col, lambda: CatanPolicy().cuda())
    assert 1 <= len(distinct) <= 2 and col.opp_index.shape == (N, 3)
    st = col.gather_rollouts()
    assert env.invalid_action_count() == 0
    assert int(col.n_act.min()) == T
    # the stored decisions belong to the central policy: re-evaluating them reproduces the stored log-probs
    f = st.obs_f[:T].reshape(T * N, -1); lists = st.lists[:T].reshape(T * N, 5, -1); lens = st.lens[:T].reshape(T * N, 5)
    with torch.no_grad():
        _, lp, _ = net.evaluate_actions(f.float(), lists, lens.long(), st.unpack_action_masks(st.action_masks.reshape(T * N, -1)),
                                        st.actions.reshape(T * N, -1))
    assert torch.allclose(lp[:, 0], st.action_log_probs.reshape(T * N), atol=2e-4)


def test_league_reference_draws_on_device(hip_lib):
    """SURVEY f2 on the device, against the REFERENCE's draws (tests/golden/league.npz = `update_opponent_policies` run on a
    fake manager, RL/ppo/update_opponent_policies.py:13-43): League.sample / League.assign on a real collector give every
    worker (group of 5 games) exactly the three snapshots the reference gave that process, policy slot by policy slot; then
    the collector's GROUPED inference (one batched forward per distinct net in play) on real observations equals PER-NET
    inference: the (action, log-prob) of every game is what the net the reference assigned to the deciding seat produces for
    that row - and not what the central policy would."""
    import golden_util as gu
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.league import League, get_prob_dist
    g = gu.load("league.npz")
    for n in g["sizes"]:
        assert np.allclose(get_prob_dist(int(n)), g[f"p_{int(n)}"], rtol=0, atol=1e-15)
    seed, nproc, npol = 0, 7, 40
    want = g[f"draw_{seed}_{nproc}_{npol}"]                          # [process][policy slot - 1] -> snapshot id
    torch.manual_seed(4)
    N = nproc * 5
    env = VecCatanEnv(N, seed=13)
    env.random_rollout(0, 400)
    base = CatanPolicy().cuda()
    lg = League(envs_per_worker=5, seed=seed)
    gen = torch.Generator(device="cuda").manual_seed(1)
    sd0 = {k: v.detach().clone() for k, v in base.state_dict().items()}
    for k in range(npol):                                            # 40 distinct snapshots (the deque keeps CPU copies)
        with torch.no_grad():
            for name, p in base.named_parameters():
                p.copy_(sd0[name] + 0.02 * (k + 1) * torch.randn(p.shape, device="cuda", generator=gen))
        lg.add(base)
    base.load_state_dict(sd0)
    col = RolloutCollector(env, base, 4, seed=3)
    distinct = lg.assign(col, lambda: CatanPolicy().cuda())
    assert np.array_equal(distinct, np.unique(want))
    per_game = distinct[col.opp_index.cpu().numpy()]                 # snapshot id per (game, opponent slot)
    assert np.array_equal(per_game, np.repeat(want, 5, axis=0)), "a worker's games must share the three snapshots the reference drew for it"
    # grouped inference == per-net inference on those groups
    f, lists, lens = env.get_obs()
    masks = env.get_action_masks()
    deciding = env.deciding_player().long()
    pol = col.policy_of_pid[torch.arange(N, device="cuda"), deciding - 1]
    assert int((pol > 0).sum()) > N // 3 and int((pol == 0).sum()) > 0
    actions, logp = col._act(f, lists, lens, masks, pol)
    nets = {}
    checked_other = 0
    for i in range(N):
        slot = int(pol[i])
        sid = -1 if slot == 0 else int(per_game[i, slot - 1])
        if sid not in nets:
            net = base
            if sid >= 0:
                net = CatanPolicy().cuda().eval()
                net.load_state_dict(lg.earlier[sid])
            nets[sid] = net
        row = (f[i:i + 1].float(), lists[i:i + 1], lens[i:i + 1].long(), masks[i:i + 1], actions[i:i + 1])
        with torch.no_grad():
            lp = nets[sid].evaluate_actions(*row)[1]
            assert abs(float(lp) - float(logp[i])) < 2e-4, (i, slot, sid, float(lp), float(logp[i]))
            if sid >= 0:
                other = base.evaluate_actions(*row)[1]
                checked_other += int(abs(float(other) - float(logp[i])) > 1e-3)
    assert checked_other > (N // 3) // 2, checked_other              # the routing matters: the central net scores those rows differently
    assert env.invalid_action_count() == 0


def test_evaluation_protocol_on_device(hip_lib):
    """run_evaluation_protocol on the HIP env (two random-init nets, sampled actions, the offline evaluator's draw cap to
    bound the run): every action legal, statistics well-formed, capped games reported as draws."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd import evaluation as ev
    import random
    torch.manual_seed(0)
    central, opp = CatanPolicy().cuda().eval(), CatanPolicy().cuda().eval()
    envs = []

    def make_env(n):
        envs.append(VecCatanEnv(n, seed=8, auto_reset=False))
        return envs[-1]

    log, summary = ev.run_evaluation_protocol(make_env, central, opp, 96, update_num=3, rng=random.Random(1), max_steps=400)
    assert envs[0].invalid_action_count() == 0
    r = log["random"]
    assert log["update"] == 3 and 0.0 <= r["policy_win_frac"] <= 1.0
    assert 400 < r["avg_game_length"] <= 401.0001 or r["policy_win_frac"] > 0      # capped games stop right after step 401
    assert 0 < r["avg_policy_decisions"] < r["avg_game_length"] and 0 <= r["avg_victory_points"] <= 12
    assert "EVALUATION (after 3 updates)" in summary


@pytest.mark.parametrize("R,K,N,bias", [(100003, 64, 192, True), (70000, 64, 25, True), (33333, 128, 64, False), (50001, 16, 48, True),
                                        (4099, 48, 16, False), (17, 64, 64, True), (65536, 128, 128, True), (20000, 8, 16, True)])
def test_linear_rows_kernel_vs_torch(hip_lib, R, K, N, bias):
    """k_linear_rows (MFMA, rows split over the grid, W in registers) against fp32 torch on the same bf16 inputs; asymmetric
    data, ragged row tails, partial k-steps and n-tiles."""
    import ctypes as C
    from settlers_of_catan_rl_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(R + K + N)
    x = (torch.randn((R, K), device="cuda", generator=g) * torch.linspace(0.5, 2.0, K, device="cuda")).to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda", generator=g) * 0.3 + torch.linspace(-0.2, 0.2, N, device="cuda")[:, None]).to(torch.bfloat16)
    b = (torch.randn((N,), device="cuda", generator=g)).to(torch.bfloat16) if bias else None
    y = torch.empty((R, N), device="cuda", dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.catan_linear_rows_supported(R, K, N)
    _lib.check(L.catan_linear_rows(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()) if bias else None,
                                   C.c_void_p(y.data_ptr()), R, K, N, st))
    ref = x.float() @ w.float().t() + (b.float() if bias else 0.0)
    assert torch.allclose(y.float(), ref, atol=2e-2 * float(ref.abs().max()) / 4 + 1e-2, rtol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,L,masked", [(1000, 256, True), (37, 8, False), (4099, 64, True)])
def test_lstm_cell_kernel_vs_torch(hip_lib, n, L, masked, dtype):
    """k_lstm_cell_fwd / _bwd against the torch.nn.LSTM cell formulas (gate order i, f, g, o) in fp64 on the same
    (rounded) inputs.  Tolerances: fp32 outputs 2e-6 abs (fast exp), gate gradients 1e-5 (fp32) / 1 bf16 ulp (bf16)."""
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(n + L)
    gx = (torch.randn(n, 4 * L, generator=g, device="cuda") * 1.5).to(dtype).requires_grad_(True)
    gh = (torch.randn(n, 4 * L, generator=g, device="cuda") * 1.5).to(dtype).requires_grad_(True)
    c0 = torch.randn(n, L, generator=g, device="cuda").requires_grad_(True)
    m = (torch.rand(n, generator=g, device="cuda") > 0.3).float() if masked else None
    wh, wc = torch.randn(n, L, generator=g, device="cuda"), torch.randn(n, L, generator=g, device="cuda")
    h, c = nn_kernels.lstm_cell(gx, gh, c0, m)
    (h * wh + c * wc).sum().backward()
    gx64, gh64, c64 = (t.detach().double().requires_grad_(True) for t in (gx, gh, c0))
    a = gx64 + gh64
    i, f, gg, o = a[:, :L], a[:, L:2 * L], a[:, 2 * L:3 * L], a[:, 3 * L:]
    cin = c64 * (m.double()[:, None] if masked else 1.0)
    c_ref = torch.sigmoid(f) * cin + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    (h_ref * wh.double() + c_ref * wc.double()).sum().backward()
    assert float((h.double() - h_ref).abs().max()) < 2e-6 and float((c.double() - c_ref).abs().max()) < 4e-6
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    for got, want in ((gx.grad, gx64.grad), (gh.grad, gh64.grad)):
        err = (got.double() - want).abs() / (1.0 + want.abs()) if dtype == torch.float32 else (got.double() - want).abs() / (want.abs() + 1e-3)
        assert float(err.max()) < tol, float(err.max())
    assert float((c0.grad.double() - c64.grad).abs().max()) < 1e-5
    assert torch.equal(gx.grad, gh.grad)


@pytest.mark.parametrize("B,K,window", [(5000, 13, True), (777, 73, False), (4096, 2, True), (1, 54, False)])
def test_masked_categorical_kernel_vs_torch(hip_lib, B, K, window):
    """k_categorical_fwd / _bwd against log_softmax(logits + log(mask)) / gather / entropy in torch (fp64): log-probs and
    entropies within 2e-6, gradients within 2e-6, arg-max actions equal, sampled actions legal and distributed like p."""
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(B + K)
    logits = (torch.randn(B, K, generator=g, device="cuda") * 2).requires_grad_(True)
    full = (torch.rand(B, 325, generator=g, device="cuda") > 0.4).float()
    full[:, 7] = 1.0                                                  # at least one legal entry per row
    mask = full[:, 5:5 + K] if window else full[:, 5:5 + K].contiguous()
    if K == 2:
        mask = full[:, 6:8]
    given = torch.multinomial(mask + 1e-9, 1, generator=g).squeeze(-1)
    wl, we = torch.randn(B, generator=g, device="cuda"), torch.randn(B, generator=g, device="cuda")
    a, lp, ent = nn_kernels.masked_categorical(logits, mask, given)
    assert torch.equal(a, given)
    (lp * wl + ent * we).sum().backward()
    z = logits.detach().double().requires_grad_(True)
    lp_all = torch.log_softmax(z + torch.log(mask.double()), -1)
    p = lp_all.exp()
    ent_ref = -(p * torch.where(p > 0, lp_all, torch.zeros_like(lp_all))).sum(-1)
    lp_ref = lp_all.gather(-1, given[:, None]).squeeze(-1)
    (lp_ref * wl.double() + ent_ref * we.double()).sum().backward()
    assert float((lp.double() - lp_ref).abs().max()) < 2e-6 and float((ent.double() - ent_ref).abs().max()) < 2e-6
    assert float((logits.grad.double() - z.grad).abs().max()) < 2e-6
    with torch.no_grad():
        a_det, lp_det, _ = nn_kernels.masked_categorical(logits, mask, None, deterministic=True)
        assert torch.equal(a_det, lp_all.argmax(-1)) and torch.allclose(lp_det.double(), lp_all.max(-1).values, atol=2e-6)
        # a given action that the mask forbids has log-prob -inf (as logits + log(0) gives)
        bad = (mask == 0).float().argmax(-1)
        has_bad = (mask == 0).any(-1)
        if bool(has_bad.any()):
            _, lp_bad, _ = nn_kernels.masked_categorical(logits, mask, bad)
            assert bool(torch.isinf(lp_bad[has_bad]).all())
        # sampling: legal, reproducible with the generator, frequencies follow p (row 0 repeated)
        g2 = torch.Generator(device="cuda").manual_seed(1)
        a_s, lp_s, _ = nn_kernels.masked_categorical(logits, mask, None, generator=g2)
        assert bool((mask.gather(-1, a_s[:, None]) > 0).all())
        assert torch.allclose(lp_s.double(), lp_all.gather(-1, a_s[:, None]).squeeze(-1), atol=2e-6)
        g3 = torch.Generator(device="cuda").manual_seed(1)
        assert torch.equal(nn_kernels.masked_categorical(logits, mask, None, generator=g3)[0], a_s)
        n = 200000
        rep_l, rep_m = logits[:1].expand(n, K).contiguous(), mask[:1].expand(n, K).contiguous()
        a_r, _, _ = nn_kernels.masked_categorical(rep_l, rep_m, None, generator=g2)
        freq = torch.bincount(a_r, minlength=K).double() / n
        assert float((freq - p[0].detach()).abs().max()) < 0.01


def test_inference_copy_acts_like_the_master_under_autocast(hip_lib):
    """policy.inference_copy: Linear / LSTM / embedding weights stored in bf16 (nothing left for autocast to cast) - the same
    decisions, values and log-probs as the fp32 master under bf16 autocast, before and after a refresh."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    torch.manual_seed(0)
    env = VecCatanEnv(512, seed=2); env.random_rollout(0, 400)
    f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
    for lstm in (False, True):
        net = CatanPolicy(include_lstm=lstm).cuda()
        inf = net.inference_copy(torch.bfloat16)
        assert inf.value_out.weight.dtype == torch.bfloat16 and inf.v_norm_1.weight.dtype == torch.float32
        assert not any(p.requires_grad for p in inf.parameters())
        kw = dict(hidden=net.initial_hidden(512), nonterminal=torch.ones(512, device="cuda")) if lstm else {}
        for rnd in range(2):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                a = net.act(f, lists, lens, masks, deterministic=True, **kw)
                b = inf.act(f, lists, lens, masks, deterministic=True, **kw)
            assert torch.equal(a[1], b[1]) and torch.allclose(a[0], b[0], atol=1e-6) and torch.allclose(a[2], b[2], atol=1e-6)
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(0.01 * torch.randn_like(p))
            inf.load_from(net)


def test_card_summary_kernel_vs_torch_formulation(hip_lib):
    """k_card_summary_fwd / _bwd (the dev-card list module per card class, one lane per list) against the same algebra in torch
    ops, which the CPU tests pin to the reference net: outputs and the gradients of every parameter it touches."""
    import torch
    from settlers_of_catan_rl_amd import nn_kernels
    from settlers_of_catan_rl_amd.policy import CatanPolicy, _card_summary
    torch.manual_seed(0)
    net = CatanPolicy().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    om = net.observation_module
    B = 5000
    g = torch.Generator(device="cuda").manual_seed(1)
    lens = torch.randint(1, 26, (B,), device="cuda", generator=g)
    lens[:700] = 1
    ids_full = torch.randint(1, 6, (B, 5, 25), device="cuda", generator=g).to(torch.int8)
    ids_full[:700, :, 0] = 0                                         # empty lists: the reference's [0]
    ids_full = ids_full * (torch.arange(25, device="cuda")[None, None, :] < lens[:, None, None]).to(torch.int8)
    w = torch.randn(B, 16, device="cuda", generator=g)
    params = [om.dev_card_embedding.weight] + list(om.played_card_mha.parameters()) + list(om.current_player_module.norm.parameters())
    res = {}
    for use_kernel in (True, False):
        saved = nn_kernels.card_summary_supported
        if not use_kernel:
            nn_kernels.card_summary_supported = lambda *a: False
        try:
            deck = torch.tensor([1] * 14 + [2] * 5 + [3] * 2 + [4] * 2 + [5] * 2, device="cuda")
            real = deck[torch.rand(B, 25, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).argsort(1)]
            real = (real * (torch.arange(25, device="cuda")[None] < lens[:, None])).to(torch.int8)
            real[:700] = 0                                                                       # (empty lists: id 0, length 1)
            # strided int8, int64, int32 with arbitrary counts (beyond the deck: the direct backward), and deck-bounded lists
            # (the per-pattern backward)
            for ids in (ids_full[:, 2], ids_full[:, 1].long().contiguous(), ids_full[:, 3].to(torch.int32), real):
                for p in params:
                    p.grad = None
                out = _card_summary(ids, lens, om.dev_card_embedding, om.played_card_mha, om.current_player_module.norm)
                (out * w).sum().backward()
                res.setdefault(use_kernel, []).append((out.detach().clone(), [p.grad.clone() for p in params]))
        finally:
            nn_kernels.card_summary_supported = saved
    for (o1, g1), (o2, g2) in zip(res[True], res[False]):
        assert o1.shape == (B, 16) and torch.allclose(o1, o2, atol=2e-5, rtol=1e-5), float((o1 - o2).abs().max())
        for a, b in zip(g1, g2):
            # fp32 sums of ~B terms in another order - and, in the kernel, in an order that atomics decide per run: 1.1e-4 ... 3.2e-4 were seen
            # over this round's suite runs (3e-4 was the bound until one run of 30 exceeded it)
            assert float((a - b).abs().max()) <= 6e-4 * max(1.0, float(b.abs().max())), float((a - b).abs().max())


def test_compact_head_evaluation_on_device(hip_lib):
    """evaluate_actions at a minibatch width where the heads run only on the rows that use them (>= compact_min_rows):
    joint log-probs, entropy and the trunk gradient equal the dense evaluation (fp32, HIP kernels on both sides)."""
    import torch
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    torch.manual_seed(0)
    n = 65536
    env = VecCatanEnv(n, seed=4); env.random_rollout(0, 700)
    f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
    net = CatanPolicy().cuda()
    with torch.no_grad():
        _, acts, _ = net.act(f, lists, lens, masks)
    assert n >= net.action_head_module.compact_min_rows
    out = {}
    for compact in (True, False):
        net.action_head_module.compact_evaluate = compact
        net.zero_grad()
        v, lp, ent = net.evaluate_actions(f, lists, lens, masks, acts)
        (lp.mean() + ent + v.mean()).backward()
        out[compact] = (lp.detach().clone(), float(ent), net.observation_module.final_layer.weight.grad.clone())
    net.action_head_module.compact_evaluate = True
    assert torch.allclose(out[True][0], out[False][0], atol=2e-4), float((out[True][0] - out[False][0]).abs().max())
    assert abs(out[True][1] - out[False][1]) < 1e-5
    assert torch.allclose(out[True][2], out[False][2], atol=1e-5, rtol=1e-3)


def test_fused_tile_encoder_forward_vs_unfused(hip_lib):
    """k_tile_encoder_fwd (the whole tile encoder in one kernel, inference) against the unfused path: both compute in bf16
    with fp32 accumulation, so each is compared with the fp32 evaluation of the same module; the fused kernel must be as
    close to it as the unfused bf16 path is.  Ragged board counts exercise partial groups of boards."""
    import torch
    from settlers_of_catan_rl_amd import nn_kernels
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    torch.manual_seed(0)
    net = CatanPolicy().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    te = net.observation_module.tile_encoder
    env = VecCatanEnv(1037, seed=3); env.random_rollout(0, 600)
    f, _, _ = env.get_obs()
    for B in (1, 7, 8, 9, 1037):
        tiles = f[:B, 18:18 + 1140].reshape(B, 19, 60)
        with torch.no_grad():
            ref32 = te(tiles.float())                                            # fp32, torch ops / fp32 kernels
            with torch.autocast("cuda", dtype=torch.bfloat16):
                assert nn_kernels.tile_encoder_supported(te, tiles)
                fused = te(tiles)
                saved = nn_kernels.tile_encoder_supported
                nn_kernels.tile_encoder_supported = lambda *a: False
                try:
                    unfused = te(tiles)
                finally:
                    nn_kernels.tile_encoder_supported = saved
        assert fused.shape == (B, 475) and fused.dtype == torch.bfloat16
        e_f = float((fused.float() - ref32).abs().max()); e_u = float((unfused.float() - ref32).abs().max())
        assert e_f <= max(2.0 * e_u, 0.06), (B, e_f, e_u)
        assert float((fused.float() - ref32).abs().mean()) <= max(2.0 * float((unfused.float() - ref32).abs().mean()), 5e-3)
    # boards are independent of their neighbours in a workgroup and of the grid size: 40 003 boards = the 1 037 repeated, bit-exact
    big = tiles.repeat(39, 1, 1)[:40003].contiguous()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        fb = te(big)
    assert torch.equal(fb, fused.repeat(39, 1)[:40003])
    # a changed parameter re-packs the weights
    with torch.no_grad():
        te.norm.bias.add_(1.0)                                                     # (after the ReLU-free LayerNorm: shifts every output)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            again = te(tiles)
    assert float((again.float() - fused.float()).abs().max()) > 0.5


def test_fused_encoder_sublayers_vs_unfused(hip_lib):
    """An encoder layer with the residual adds / the FFN's ReLU fused into the row-kernel products (catan_linear_rows_fused,
    modes 1-3) and the residual stream's gradient added inside the LayerNorm backward (catan_layer_norm_bwd_res) against the
    same layer with the separate elementwise ops and autograd's own gradient accumulation: the fused epilogues act on the bf16-rounded
    product exactly as the separate ops do, so outputs and every gradient are identical."""
    import torch
    from settlers_of_catan_rl_amd import nn_kernels
    from settlers_of_catan_rl_amd.policy import _EncoderLayer
    torch.manual_seed(0)
    layer = _EncoderLayer(64, 4).cuda()
    B = 16000                                                                      # x 19 tokens = 304 000 rows (>= the row kernels' minimum)
    x = torch.randn(B, 19, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(B, 19, 64, device="cuda").to(torch.bfloat16)

    def run():
        for p in layer.parameters():
            p.grad = None
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = layer(x)
        y.backward(g)
        return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in layer.parameters()]

    assert nn_kernels.fused_sublayer_supported(x, 64, 128)
    y1, dx1, gp1 = run()
    assert nn_kernels.pre_norm_supported(x, layer.sublayers[0].norm)               # x's gradient formed in the LayerNorm backward
    saved = nn_kernels.fused_sublayer_supported, nn_kernels.pre_norm_supported
    nn_kernels.fused_sublayer_supported = lambda *a, **k: False
    nn_kernels.pre_norm_supported = lambda *a, **k: False
    try:
        y0, dx0, gp0 = run()
    finally:
        nn_kernels.fused_sublayer_supported, nn_kernels.pre_norm_supported = saved
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))      # (fp32 atomics: order of the row chunks)
    # and the kernel's argument checks
    import ctypes as C
    L = hip_lib
    t = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    assert L.catan_linear_rows_fused(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), None, C.c_void_p(t.data_ptr()), 64, 64, 64, None, 2, None) != 0
    assert L.catan_linear_rows_fused(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), None, C.c_void_p(t.data_ptr()), 64, 64, 64, None, 4, None) != 0


def test_collector_graphed_act_uses_current_weights(hip_lib):
    """The collector's policy pass as a captured hipGraph (self-play, feed-forward net, >= 8 192 games): legal actions only,
    and the stored log-probs are those of the CURRENT central weights - also after an optimiser-style change of every
    parameter (a replay does not run the host code that packs the fused tile encoder's parameters; the collector must)."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    torch.manual_seed(0)
    N, T = 8192, 3
    env = VecCatanEnv(N, seed=9)
    env.random_rollout(0, 300)
    net = CatanPolicy().cuda()
    col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
    assert col.graph_act
    for rnd in range(2):
        st = col.gather_rollouts()
        assert col._graphed is not None and not col._graphed.failed and N in col._graphed.graphs
        assert env.invalid_action_count() == 0
        f = st.obs_f[:T].reshape(T * N, -1); lists = st.lists[:T].reshape(T * N, 5, -1); lens = st.lens[:T].reshape(T * N, 5).long()
        masks = st.unpack_action_masks(st.action_masks[:T].reshape(T * N, -1))
        acts = st.actions[:T].reshape(T * N, -1)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            _, lp, _ = col._shadow.evaluate_actions(f, lists, lens, masks, acts)
        stored = st.action_log_probs[:T].reshape(T * N)
        err = (lp.float().reshape(-1) - stored).abs()
        assert float(err.max()) < 0.05 and float(err.mean()) < 2e-3, (rnd, float(err.max()), float(err.mean()))
        with torch.no_grad():                                  # "an update": every parameter moves
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
        col.after_rollouts()
    # the eager collector on the same states emits from the same distribution (spot check: same mean log-prob within noise)
    col2 = RolloutCollector(VecCatanEnv(N, seed=9), net, T, seed=1, autocast_dtype=torch.bfloat16, graph_act=False)
    assert not col2.graph_act


def test_unpack_action_masks_kernel_vs_torch(hip_lib):
    """RolloutStorage.unpack_action_masks on the GPU (catan_expand_masks: packed int32 [.., 11] -> float32 [.., 325]) against the
    torch bit arithmetic it replaces, on real masks and on random bit patterns (incl. bit 31 of every word)."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.rollout import RolloutStorage, pack_action_masks
    env = VecCatanEnv(1000, seed=4); env.random_rollout(0, 700)
    m = env.get_action_masks()
    st = RolloutStorage(2, 8, "cuda")
    packed = pack_action_masks(m)
    assert torch.equal(st.unpack_action_masks(packed), m)
    rnd = torch.randint(-2 ** 31, 2 ** 31 - 1, (3, 77, 11), dtype=torch.int64, device="cuda").int()
    bits = (rnd[..., None] >> torch.arange(32, device="cuda", dtype=torch.int32)) & 1
    ref = bits.reshape(3, 77, 352)[..., :spec.MASK_WORDS].float()
    assert torch.equal(st.unpack_action_masks(rnd), ref)
    assert torch.equal(st.unpack_action_masks(rnd.cpu()), ref.cpu())              # (the torch path, CPU tensors)


def test_masked_row_store_and_packed_masks(hip_lib):
    """catan_masked_row_store (dst[t[r], r] = src[r] where sel[r]) against the torch indexing it replaces, for 4- / 2- / 1-byte
    aligned rows; catan_masks_packed_copy against packing the float masks."""
    import ctypes as C
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.rollout import pack_action_masks
    L = hip_lib
    g = torch.Generator(device="cuda").manual_seed(5)
    N, S = 3001, 7
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shape, dt in (((1787,), torch.bfloat16), ((5, 25), torch.int8), ((16,), torch.float32), ((3,), torch.int8)):
        dst = torch.zeros((S, N) + shape, device="cuda").to(dt)
        dst.copy_(torch.randint(0, 100, dst.shape, device="cuda", generator=g))
        ref = dst.clone()
        src = torch.randint(0, 100, (N,) + shape, device="cuda", generator=g).to(dt)
        t = torch.randint(0, S, (N,), device="cuda", generator=g)
        sel = torch.rand(N, device="cuda", generator=g) < 0.3
        idx = sel.nonzero(as_tuple=True)[0]
        ref[t[idx], idx] = src[idx]
        sel8 = sel.to(torch.uint8)
        rb = src[0].numel() * src.element_size()
        assert L.catan_masked_row_store(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(sel8.data_ptr()),
                                        N, rb, dst.stride(0) * dst.element_size(), st) == 0
        assert torch.equal(dst, ref), (shape, dt)
    assert L.catan_masked_row_store(None, None, None, None, 1, 1, 1, st) != 0
    env = VecCatanEnv(2000, seed=8); env.random_rollout(0, 900)
    assert torch.equal(env.get_action_masks_packed(), pack_action_masks(env.get_action_masks()))


def test_forked_inference_streams_change_nothing(hip_lib):
    """policy._Branches: the independent chains of an inference pass on side streams - same actions, values and log-probs as
    the single-stream pass (same generator state), eagerly and inside a captured hipGraph."""
    from settlers_of_catan_rl_amd import policy as P
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.forward_search import GraphedAct
    torch.manual_seed(0)
    env = VecCatanEnv(4096, seed=12); env.random_rollout(0, 800)
    f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
    net = P.CatanPolicy().cuda().inference_copy(torch.bfloat16)

    def act(seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return net.act(f, lists, lens, masks, generator=g)

    a1 = act(3)
    saved = P._Branches.enabled
    P._Branches.enabled = False
    try:
        a0 = act(3)
    finally:
        P._Branches.enabled = saved
    assert torch.equal(a1[1], a0[1]) and torch.equal(a1[0], a0[0]) and torch.equal(a1[2], a0[2])
    gen = torch.Generator(device="cuda").manual_seed(11)
    ga = GraphedAct(net, buckets=(4096,), autocast_dtype=torch.bfloat16, generator=gen)
    v, a, lp = ga(f, lists, lens, masks, with_logp=True)
    assert not ga.failed and 4096 in ga.graphs
    # the replayed actions are legal and their log-probs are the net's own evaluation of them
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, lp_e, _ = net.evaluate_actions(f, lists, lens, masks, a)
    assert float((lp_e.float().reshape(-1) - lp.float().reshape(-1)).abs().max()) < 0.05
    assert (masks[:, :13].gather(1, a[:, :1]) == 1).all()


def test_fused_head_kernel_vs_unfused_heads(hip_lib):
    """csrc/catan_heads.hip (one kernel per head evaluation: conditioning add, LayerNorm + ReLU, 128 x 128, 128 x K, masked
    categorical) against the unfused bf16 chain of the same heads: (1) every head alone, arg-max - the same action and a
    log-prob within 0.02 on all but a handful of near-tie rows; (2) the whole autoregressive pass, arg-max and sampled with the
    same uniforms; (3) the sampled actions' log-probs equal the net's own evaluation of them; every sampled action is legal."""
    from settlers_of_catan_rl_amd import policy as P, nn_kernels
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    torch.manual_seed(0)
    B = 4096 + 37                                   # not a multiple of the 256 rows a workgroup takes
    env = VecCatanEnv(B, seed=21); env.random_rollout(0, 900)
    f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
    net = P.CatanPolicy().cuda()
    with torch.no_grad():
        for p in net.parameters():                  # decisive heads (the default init has near-uniform output layers)
            p.add_(0.05 * torch.randn_like(p))
    net = net.inference_copy(torch.bfloat16)
    ahm = net.action_head_module
    # (1) head by head on random bf16 trunk products and random conditioning columns
    g = torch.Generator(device="cuda").manual_seed(5)
    pre_all = (torch.randn(B, 12 * 128, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    for i, head in enumerate(ahm.action_heads):
        K = head.distribution.linear.weight.shape[0]
        e = head.mlp_1.weight.shape[1] - ahm.D
        cond = None if e == 0 else torch.randint(0, 3, (B, e), device="cuda", generator=g).float()
        mask = (torch.rand(B, K, device="cuda", generator=g) < 0.6).float()
        mask[:, 0] = 1.0
        pre = pre_all[:, 128 * i:128 * (i + 1)]
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            a1, lp1 = nn_kernels.head_sample(head, ahm.D, pre, cond, mask, deterministic=True)
            logits = head.logits(pre, cond, None) if i != 5 else head.logits(pre, cond, None)
            a0, lp0, _ = P._categorical(logits, mask, None, True, None)
        same = a1 == a0
        assert float(same.float().mean()) > 0.995, (i, float(same.float().mean()))
        assert float((lp1[same] - lp0[same]).abs().max()) < 0.03, (i, float((lp1[same] - lp0[same]).abs().max()))
        assert bool((mask.gather(1, a1[:, None]) == 1).all()), i
        u = torch.rand(B, device="cuda", generator=g)

        class _U(object):                            # hands the same uniforms to both paths
            def take(self, rows): return u
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            wts, vec = nn_kernels.head_pack(head, ahm.D)
        act = torch.empty(B, dtype=torch.int64, device="cuda"); lp = torch.empty(B, device="cuda")
        import ctypes as C
        from settlers_of_catan_rl_amd import _lib
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().catan_head_fwd(p(pre), pre.stride(0), p(cond), cond.stride(0) if cond is not None else 0, e, p(wts), p(vec), float(head.norm.eps), K,
                                             p(mask), mask.stride(0), p(u), p(act), p(lp), B, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        from settlers_of_catan_rl_amd.nn_kernels import _MaskedCategorical
        a0s, lp0s, _ = _MaskedCategorical.apply(logits, mask, None, u)
        same = act == a0s
        assert float(same.float().mean()) > 0.99, (i, float(same.float().mean()))
        assert bool((mask.gather(1, act[:, None]) == 1).all()), i
    # (2) the whole pass
    def act_pass(fused, deterministic, seed=3):
        nn_kernels.fused_heads_enabled = fused
        try:
            gg = torch.Generator(device="cuda").manual_seed(seed)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return net.act(f, lists, lens, masks, deterministic=deterministic, generator=gg)
        finally:
            nn_kernels.fused_heads_enabled = True
    v1, a1, lp1 = act_pass(True, True)
    v0, a0, lp0 = act_pass(False, True)
    assert torch.equal(v1, v0)
    assert float((a1[:, 0] == a0[:, 0]).float().mean()) > 0.995
    same = (a1 == a0).all(1)
    assert float(same.float().mean()) > 0.95, float(same.float().mean())
    assert float((lp1[same] - lp0[same]).abs().max()) < 0.05
    v1, a1, lp1 = act_pass(True, False)
    v0, a0, lp0 = act_pass(False, False)
    assert float((a1[:, 0] == a0[:, 0]).float().mean()) > 0.99          # same uniforms: the same type nearly everywhere
    # (3) legal, and the log-probs are the net's own evaluation of the sampled actions
    assert (masks[:, :13].gather(1, a1[:, :1]) == 1).all()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, lp_e, _ = net.evaluate_actions(f, lists, lens, masks, a1)
    assert float((lp_e.float().reshape(-1) - lp1.float().reshape(-1)).abs().max()) < 0.06
    r, d = env.step(a1.to(torch.int32))
    assert env.invalid_action_count() == 0


def test_card_summary_pattern_table_lookup(hip_lib):
    """Inference form of the dev-card list module (k_card_summary_lookup: count pattern -> a row of the 4 860-pattern table) against
    the direct kernel on real lists, after a weight change too (the table is rebuilt in place), and on lists whose counts fall
    outside the deck (evaluated directly)."""
    from settlers_of_catan_rl_amd import policy as P, nn_kernels
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    torch.manual_seed(0)
    env = VecCatanEnv(4096, seed=7); env.random_rollout(0, 1500)
    f, lists, lens = env.get_obs(); lens = lens.long()
    net = P.CatanPolicy().cuda()
    om = net.observation_module
    for rnd in range(2):
        for li, (mha, norm) in enumerate(((om.played_card_mha, om.current_player_module.norm), (om.hidden_card_mha, om.current_player_module.norm),
                                          (om.played_card_mha, om.other_players_module.norm))):
            ids, ln = lists[:, li], lens[:, li]
            direct = P._card_summary(ids, ln, om.dev_card_embedding, mha, norm)                  # grad enabled: k_card_summary_fwd
            with torch.no_grad():
                looked = P._card_summary(ids, ln, om.dev_card_embedding, mha, norm)
            assert torch.allclose(looked, direct.detach(), rtol=1e-6, atol=1e-6), float((looked - direct).abs().max())
        with torch.no_grad():
            for p in om.parameters():
                p.add_(0.05 * torch.randn_like(p))
    odd = torch.randint(0, 6, (512, 25), device="cuda", dtype=torch.int32)                       # 25 random ids: counts outside the deck
    ln = torch.full((512,), 25, device="cuda")
    direct = P._card_summary(odd, ln, om.dev_card_embedding, om.played_card_mha, om.current_player_module.norm)
    with torch.no_grad():
        looked = P._card_summary(odd, ln, om.dev_card_embedding, om.played_card_mha, om.current_player_module.norm)
    assert torch.allclose(looked, direct.detach(), rtol=1e-5, atol=1e-5)


def test_value_reevaluation_encodes_distinct_boards_only(hip_lib):
    """PPOTrainer.compute_values runs the tile encoder once per DISTINCT board of a game's stored observations (consecutive
    observations of the active seat mostly show the same board): the same values, bit for bit, as encoding every row; and the
    run bookkeeping: the first row of every run really starts a new board, every other row repeats its predecessor's."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    N, T = 2048, 24
    env = VecCatanEnv(N, seed=15); env.random_rollout(0, 900)
    net = CatanPolicy().cuda()
    col = RolloutCollector(env, net, T, seed=2, autocast_dtype=torch.bfloat16)
    tr = PPOTrainer(net, PPOConfig(value_chunk=8192), autocast_dtype=torch.bfloat16, seed=0)
    for rnd in range(2):
        st = col.gather_rollouts()
        first_rows, board_of_row = tr.board_runs(st)
        o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
        tiles = st.obs_f.reshape((T + 1) * N, -1)[:, o:o + 1140]
        assert torch.equal(tiles[first_rows[board_of_row]], tiles)                 # every row's board is the one it points to
        frac = first_rows.numel() / tiles.shape[0]
        assert 0.02 < frac < 0.6, frac                                             # most rows repeat a board
        tr.dedupe_boards = True
        v1 = tr.compute_values(st)
        tr.dedupe_boards = False
        v0 = tr.compute_values(st)
        tr.dedupe_boards = True
        assert torch.equal(v1, v0)
        col.after_rollouts()


def test_chained_heads_equal_the_glued_heads(hip_lib):
    """nn_kernels.heads_chain (the twelve heads' glue inside the fused head kernels: type-conditional mask rows, conditioning
    columns, log-prob masks, the trade heads' lists, condition_on_action_type) against the same kernels with the glue as torch ops:
    the same uniforms give the same 18 action columns and the same joint log-prob, arg-max and sampled, with and without forced
    types; the sampled actions are legal in the env."""
    from settlers_of_catan_rl_amd import policy as P, nn_kernels
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    torch.manual_seed(0)
    for B in (8192 + 5, 49152 + 333):           # (the narrow and the wide configuration of the head kernel)
        _chained_vs_glued(P, nn_kernels, VecCatanEnv, B)


def _chained_vs_glued(P, nn_kernels, VecCatanEnv, B):
    env = VecCatanEnv(B, seed=33); env.random_rollout(0, 1100)
    f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
    net = P.CatanPolicy().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    net = net.inference_copy(torch.bfloat16)
    legal_types = masks[:, :13] > 0
    forced = torch.where(torch.rand(B, device="cuda") < 0.5, torch.multinomial(legal_types.float(), 1).squeeze(1), torch.full((B,), -1, device="cuda"))

    def act_pass(chained, deterministic, cond=None, seed=3):
        nn_kernels.chained_heads_enabled = chained
        try:
            gg = torch.Generator(device="cuda").manual_seed(seed)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return net.act(f, lists, lens, masks, deterministic=deterministic, generator=gg, condition_on_action_type=cond)
        finally:
            nn_kernels.chained_heads_enabled = True
    kinds = set()
    for deterministic in (True, False):
        for cond in (None, forced):
            v1, a1, lp1 = act_pass(True, deterministic, cond)
            v0, a0, lp0 = act_pass(False, deterministic, cond)
            assert torch.equal(v1, v0)
            same = (a1 == a0).all(1)
            assert float(same.float().mean()) > 0.999, (deterministic, cond is not None, float(same.float().mean()))
            assert torch.isfinite(lp1).all()
            assert float((lp1[same] - lp0[same]).abs().max()) < 2e-3, float((lp1[same] - lp0[same]).abs().max())
            if cond is not None:
                assert bool((a1[:, 0][cond >= 0] == cond[cond >= 0]).all())
            kinds |= set(a1[:, 0].tolist())
    assert len(kinds) >= 11, kinds
    v, a, lp = act_pass(True, False, None, seed=9)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, lp_e, _ = net.evaluate_actions(f, lists, lens, masks, a)
    assert float((lp_e.float().reshape(-1) - lp.float().reshape(-1)).abs().max()) < 0.06
    env.step(a.to(torch.int32))
    assert env.invalid_action_count() == 0


def test_minibatch_steps_encode_distinct_boards_only(hip_lib):
    """PPOTrainer.update with the tile encoder run once per DISTINCT board of a minibatch (forward: spread by index; backward: the
    rows' gradients summed per board) against the same update encoding every row: the bookkeeping is exact (every row's board is
    the one it is mapped to) and the update is the same function - losses equal within bf16 noise, parameters after the update
    close; fp32: equal within 1e-5."""
    import copy
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    N, T = 2048, 24
    env = VecCatanEnv(N, seed=19); env.random_rollout(0, 900)
    net0 = CatanPolicy().cuda()
    for ac, tol_loss, tol_par in ((None, 1e-5, 2e-5), (torch.bfloat16, 3e-3, 2e-3)):
        col = RolloutCollector(env, net0, T, seed=2, autocast_dtype=ac)
        st = col.gather_rollouts()
        res = {}
        for dedupe in (True, False):
            net = copy.deepcopy(net0)
            tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=2), autocast_dtype=ac, seed=5)
            tr.dedupe_boards = dedupe
            if dedupe:
                first_rows, board_of_row = tr.board_runs(st)
                perm = torch.randperm(T * N, device="cuda")
                o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
                tiles = st.obs_f.reshape((T + 1) * N, -1)[:, o:o + 1140]
                mbs = T * N // 2
                for k, (uq, inv, order, start) in enumerate(tr.minibatch_boards(board_of_row[:T * N], perm, 2, mbs)):
                    idx = perm[k * mbs:(k + 1) * mbs]
                    assert torch.equal(tiles[first_rows[uq]][inv], tiles[idx])          # every row gets exactly its own board
                    assert uq.numel() < 0.97 * mbs and torch.equal(torch.unique(uq), uq)
                    # the rows sorted by board, and where each board's run starts: run u holds exactly the rows of board u
                    assert start.numel() == uq.numel() + 1 and int(start[0]) == 0 and int(start[-1]) == mbs and bool((start[1:] > start[:-1]).all())
                    assert torch.equal(torch.sort(order)[0], torch.arange(mbs, device="cuda"))
                    run_of = torch.repeat_interleave(torch.arange(uq.numel(), device="cuda"), start[1:] - start[:-1])
                    assert torch.equal(inv[order], run_of)
            losses = tr.update(st)
            res[dedupe] = (losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
        for a, b in zip(res[True][0], res[False][0]):
            assert abs(a - b) < tol_loss * max(1.0, abs(b)), (ac, res[True][0], res[False][0])
        assert float((res[True][1] - res[False][1]).abs().max()) < tol_par, (ac, float((res[True][1] - res[False][1]).abs().max()))
        col.after_rollouts()


def test_fused_collector_bookkeeping_equals_the_tensor_form(hip_lib):
    """catan_collector_pre / _post (one lane per game; game_manager.py:91-136) against the tensor-operation form of the same
    bookkeeping (RolloutCollector.fused_bookkeeping = False): two collectors on identically seeded envs, same net, same sampling
    generator - every rollout tensor, counter and flag identical, over two gather calls with after_rollouts between them, with
    games finishing (dense terminal rewards) and games frozen at T + 1 observations while others still play."""
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    torch.manual_seed(0)
    N, T = 1024, 6
    net = CatanPolicy().cuda()
    cols = []
    for fused in (True, False):
        env = VecCatanEnv(N, seed=5, dense_reward=True)
        env.random_rollout(0, 1500)                      # late game: some games end inside the rollout
        col = RolloutCollector(env, net, T, seed=3, graph_act=False, deferred_window=0)    # (catan_step in both: the sampled actions depend on the iteration a game plays in)
        col.fused_bookkeeping = fused
        cols.append(col)
    for rnd in range(2):
        for c in cols:                   # (the two forms notice "every game is frozen" a different number of no-op iterations late: the
            c.sample_gen.manual_seed(40 + rnd)   # policy passes of those iterations draw from the generator too)
        sts = [c.gather_rollouts() for c in cols]
        a, b = cols
        assert a.iters == b.iters and sts[0].games_complete == sts[1].games_complete, (a.iters, b.iters, sts[0].games_complete, sts[1].games_complete)
        if rnd == 0:
            assert sts[0].games_complete > 0
        for name in ("obs_f", "lists", "lens", "actions", "action_log_probs", "action_masks", "rewards", "masks"):
            x, y = getattr(sts[0], name), getattr(sts[1], name)
            assert torch.equal(x, y), (rnd, name, int((x != y).sum()))
        for name in ("n_obs", "n_msk", "n_act", "n_rew", "racc", "done_since", "pending_obs"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (rnd, name)
        assert a.env.invalid_action_count() == 0
        for c in cols:
            c.after_rollouts()


def test_fused_tile_encoder_training_forward_vs_unfused(hip_lib, monkeypatch):
    """The tile encoder's TRAINING forward as the one fused kernel that also leaves what the backward kernels read
    (catan_tile_encoder_fwd_train + nn_kernels._TileEncoderTrain) against the unfused training path (one kernel per sub-layer,
    autograd between them): same output within bf16 rounding, and every parameter's gradient as close to the fp32 gradient of
    the same module as the unfused bf16 path's is.  1 037 boards take the library fall-backs of the row products, 14 518 boards
    (275 842 token rows) the row kernels; ragged counts exercise partial groups of boards."""
    import torch
    from settlers_of_catan_rl_amd import nn_kernels
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    torch.manual_seed(0)
    net = CatanPolicy().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    te = net.observation_module.tile_encoder
    env = VecCatanEnv(1037, seed=3); env.random_rollout(0, 600)
    f, _, _ = env.get_obs()
    base = f[:, 18:18 + 1140].reshape(1037, 19, 60)
    names = [n for n, _ in te.named_parameters()]

    def run(tiles, mode):
        for p in te.parameters():
            p.grad = None
        gout = torch.Generator(device="cuda").manual_seed(5)
        if mode == "fp32":
            out = te(tiles.float())
        else:
            monkeypatch.setenv("CATAN_TE_TRAIN_UNFUSED", "1" if mode == "unfused" else "0")
            monkeypatch.setenv("CATAN_TE_BWD_UNFUSED", "1" if mode == "fused, backward in separate steps" else "0")
            monkeypatch.setenv("CATAN_TE_BWD_W", "0" if mode == "fused, weight gradients in their own kernels" else "1")
            monkeypatch.setenv("CATAN_TE_BWD_OP", "0" if mode == "fused, out-projection backward in its own kernels" else "1")
            monkeypatch.setenv("CATAN_TE_RECOMPUTE_N", {"fused, LayerNorm outputs stored": "0", "fused, LayerNorm-2 outputs stored": "1"}.get(mode, "2"))
            monkeypatch.setenv("CATAN_TE_RECOMPUTE_H", "1" if mode == "fused, hidden FFN activation recomputed" else "0")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                assert nn_kernels.tile_encoder_train_supported(te, tiles) == (mode != "unfused")
                out = te(tiles)
        w = torch.randn(out.shape, device="cuda", generator=gout)
        (out.float() * w).sum().backward()
        return out.detach().float(), {n: p.grad.detach().float().clone() for n, p in te.named_parameters()}

    for B in (9, 1037, 14518):
        tiles = base.repeat((B + 1036) // 1037, 1, 1)[:B].contiguous()
        o32, g32 = run(tiles, "fp32")
        ou, gu = run(tiles, "unfused")
        of, gf = run(tiles, "fused")
        oc, gc = run(tiles, "fused, backward in separate steps")      # the pointwise sub-layer's backward as three kernels instead of k_ffn_bwd_dx
        ow, gw_ = run(tiles, "fused, weight gradients in their own kernels")   # k_ffn_bwd_dx + two catan_linear_wgrad instead of k_ffn_bwd_w
        oo, go_ = run(tiles, "fused, out-projection backward in its own kernels")   # k_ffn_bwd_w<false> + row product + catan_linear_wgrad
        # the default backward recomputes the LayerNorm outputs n1 / n2 from their inputs (k_qkv_bwd_w<true>, k_ffn_bwd_w<., true>);
        # here the forward stores both and the passes read them / stores n2 only
        on, gn = run(tiles, "fused, LayerNorm outputs stored")
        on2, gn2 = run(tiles, "fused, LayerNorm-2 outputs stored")
        # ... and with the FFN's hidden activation h = relu(n2 W1^T + b1) recomputed instead of stored (k_ffn_bwd_w<true, true, true>; off by
        # default: measured at parity).  The recomputed h is the forward kernel's own arithmetic on the same bf16 inputs, so the two
        # backward passes see the same h up to MFMA summation order
        oh, gh = run(tiles, "fused, hidden FFN activation recomputed")
        assert torch.equal(oh, of)
        for n in names:
            scale = float(g32[n].norm()) + 1e-3 * max(float(x.norm()) for x in g32.values())
            assert float((gf[n] - gh[n]).norm()) / scale <= 0.01, (B, n, float((gf[n] - gh[n]).norm()) / scale)
        assert of.shape == (B, 475) and torch.equal(oc, of) and torch.equal(ow, of) and torch.equal(oo, of) and torch.equal(on, of) and torch.equal(on2, of)
        for n in names:
            scale = float(g32[n].norm()) + 1e-3 * max(float(x.norm()) for x in g32.values())
            assert float((gf[n] - go_[n]).norm()) / scale <= 0.02, (B, n, float((gf[n] - go_[n]).norm()) / scale)
            assert float((gf[n] - gn[n]).norm()) / scale <= 0.01, (B, n, float((gf[n] - gn[n]).norm()) / scale)
            assert float((gf[n] - gn2[n]).norm()) / scale <= 0.01, (B, n, float((gf[n] - gn2[n]).norm()) / scale)
        for n in names:
            scale = float(g32[n].norm()) + 1e-3 * max(float(x.norm()) for x in g32.values())
            assert float((gf[n] - gc[n]).norm()) / scale <= 0.02, (B, n, float((gf[n] - gc[n]).norm()) / scale)
            assert float((gf[n] - gw_[n]).norm()) / scale <= 0.02, (B, n, float((gf[n] - gw_[n]).norm()) / scale)
        e_f, e_u = float((of - o32).abs().max()), float((ou - o32).abs().max())
        assert e_f <= max(2.0 * e_u, 0.06), (B, e_f, e_u)
        assert set(gf) == set(names)
        for n in names:
            scale = float(g32[n].norm()) + 1e-3 * max(float(x.norm()) for x in g32.values())
            d_f, d_u = float((gf[n] - g32[n]).norm()) / scale, float((gu[n] - g32[n]).norm()) / scale
            assert d_f <= max(2.0 * d_u, 0.05), (B, n, d_f, d_u)


def test_concat_rows_kernel(hip_lib):
    """nn_kernels.concat_rows (catan_concat_rows) against torch.cat, values and gradients (the trunk input's 480 | 128 | 384 columns)."""
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(5)
    for rows, widths in ((4099, (480, 128, 384)), (1, (8, 8)), (70001, (64, 8, 16, 40))):
        parts = [torch.randn(rows, w, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True) for w in widths]
        y = nn_kernels.concat_rows(parts)
        assert type(y.grad_fn).__name__ == "_ConcatRowsBackward" and torch.equal(y, torch.cat(parts, -1))
        dy = torch.randn_like(y)
        gs = torch.autograd.grad(y, parts, dy)
        c = 0
        for w, gk in zip(widths, gs):
            assert torch.equal(gk, dy[:, c:c + w]); c += w
    odd = [torch.randn(5, 12, device="cuda").to(torch.bfloat16), torch.randn(5, 8, device="cuda").to(torch.bfloat16)]      # 24-byte rows: torch.cat
    assert torch.equal(nn_kernels.concat_rows(odd), torch.cat(odd, -1))


def test_gather_of_permutation_ranges_backward(hip_lib):
    """nn_kernels.gather_ranges (forward: the row gather; backward: catan_scatter_rows_ranges) against plain indexing under autograd:
    the heads' row lists are ranges of one permutation, a row sits in 0..3 of them.  Rows with one contribution are bit-equal; sums
    agree with the fp32 sum of the bf16 rows rounded once."""
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(3)
    for n, W in ((1000, 512), (40961, 512), (777, 128)):
        perm = torch.randperm(n, device="cuda", generator=g)
        cuts = sorted(torch.randint(0, n, (6,), generator=torch.Generator().manual_seed(n)).tolist())
        spans = [(cuts[0], cuts[2]), (cuts[1], cuts[3]), (cuts[1], cuts[2]), (cuts[4], cuts[5]), (cuts[5], cuts[5]), (0, cuts[0] // 2)]
        ranges, off = [], 0
        for a, b in spans:
            if b > a:
                ranges.append((a, b, off)); off += b - a
        idx = torch.cat([perm[a:b] for a, b, _ in ranges])
        src = torch.randn(n, W, device="cuda", generator=g).to(torch.bfloat16)
        dy = torch.randn(idx.numel(), W, device="cuda", generator=g).to(torch.bfloat16)
        a1 = src.clone().requires_grad_(True)
        y1 = nn_kernels.gather_ranges(a1, perm, idx, ranges)
        assert type(y1.grad_fn).__name__ == "_GatherRangesBackward" and torch.equal(y1, src[idx])
        g1, = torch.autograd.grad(y1, a1, dy)
        ref = torch.zeros(n, W, device="cuda", dtype=torch.float32).index_add_(0, idx, dy.float())
        cnt = torch.bincount(idx, minlength=n)
        assert int(cnt.max()) == 3 or n < 2000
        assert torch.equal(g1, ref.to(torch.bfloat16)), (n, W)
        assert not bool(g1[cnt == 0].any())
        # the source's two other consumers through the same node: their gradients join in the same pass (fp32 sum, rounded once)
        a2 = src.clone().requires_grad_(True)
        x0, x1, y2 = nn_kernels.fanout_gather_ranges(a2, perm, idx, ranges)
        assert type(y2.grad_fn).__name__ == "_FanOutGatherRangesBackward" and torch.equal(y2, src[idx]) and torch.equal(x0, src) and torch.equal(x1, src)
        e0 = torch.randn(n, W, device="cuda", generator=g).to(torch.bfloat16); e1 = torch.randn(n, W, device="cuda", generator=g).to(torch.bfloat16)
        g3, = torch.autograd.grad([x0, x1, y2], a2, [e0, e1, dy])
        ref3 = ref + e0.float() + e1.float()
        assert float((g3.float() - ref3).abs().max()) <= 0.04 and float((g3 != ref3.to(torch.bfloat16)).float().mean()) < 1e-3
        x0, x1, y2 = nn_kernels.fanout_gather_ranges(a2, perm, idx, ranges)
        g4, = torch.autograd.grad([x0.float().sum() * 0 + x1.float().mul(e1.float()).sum() + y2.float().sum() * 0], a2)      # only one consumer has a non-zero gradient
        assert torch.allclose(g4.float(), e1.float(), atol=1e-2)
        # a column window of a wider gradient is read in place
        wide = torch.randn(idx.numel(), W + 64, device="cuda", generator=g).to(torch.bfloat16)
        g2, = torch.autograd.grad(nn_kernels.gather_ranges(a1, perm, idx, ranges), a1, wide[:, 32:32 + W])
        assert torch.equal(g2, torch.zeros(n, W, device="cuda").index_add_(0, idx, wide[:, 32:32 + W].float()).to(torch.bfloat16))


def test_row_gather_expand_and_segment_sum_kernels(hip_lib):
    """catan_gather_rows / catan_expand_rows / catan_segment_sum_rows against torch indexing: 2-byte aligned rows of odd word counts
    taken out of a wider matrix (the rollout's 3 574-byte bf16 rows), into a column window of a wider destination; the per-board
    sums of the backward against index_add in fp32."""
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(0)
    src = torch.randn(5000, 1787, device="cuda", generator=g).to(torch.bfloat16)
    idx = torch.randint(0, 5000, (3001,), device="cuda", generator=g)
    for c0, c1 in ((18, 1158), (0, 18), (1158, 1787), (1, 1786), (0, 1787)):
        got = nn_kernels.gather_rows(src[:, c0:c1], idx)
        assert got.is_contiguous() and torch.equal(got, src[:, c0:c1][idx]), (c0, c1)
    out = torch.zeros(3001, 1787, dtype=torch.bfloat16, device="cuda")
    nn_kernels.gather_rows(src[:, 1158:], idx, out=out[:, 1158:])
    assert torch.equal(out[:, 1158:], src[:, 1158:][idx]) and not bool(out[:, :1158].any())
    i8 = torch.randint(-3, 9, (5000, 5, 25), device="cuda", generator=g).to(torch.int8)[:, :, :24]      # rows of 5 x 24 bytes with pitch 125: not contiguous rows
    assert torch.equal(nn_kernels.gather_rows(i8, idx), i8[idx])                                          # (falls back)
    i8c = torch.randint(-3, 9, (5000, 5, 26), device="cuda", generator=g).to(torch.int8)
    assert torch.equal(nn_kernels.gather_rows(i8c, idx), i8c[idx])
    f32 = torch.randn(5000, 1787, device="cuda", generator=g)
    assert torch.equal(nn_kernels.gather_rows(f32[:, 18:1158], idx), f32[:, 18:1158][idx])
    # expand + per-board sums
    U, B, W = 700, 4000, 512
    inv = torch.randint(0, U, (B,), device="cuda", generator=g)
    inv[:U] = torch.arange(U, device="cuda")                                # every board shown by at least one row
    order = torch.argsort(inv, stable=True)
    counts = torch.bincount(inv, minlength=U)
    start = torch.cat((torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(counts, 0)))
    srcu = torch.randn(U, W, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    y = nn_kernels.expand_rows(srcu, inv, order, start)
    assert type(y.grad_fn).__name__ == "_ExpandRowsBackward" and torch.equal(y, srcu[inv])
    dy = torch.randn(B, W, device="cuda", generator=g).to(torch.bfloat16)
    (gsrc,) = torch.autograd.grad(y, srcu, dy)
    ref = torch.zeros(U, W, device="cuda").index_add_(0, inv, dy.float())
    assert gsrc.dtype == torch.bfloat16 and float((gsrc.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert torch.equal(gsrc[counts == 1], dy[order[start[:-1][counts == 1]]])  # single-row boards: the row's own bits


def test_weight_images_change_nothing(hip_lib):
    """nn_kernels.weight_images (persistent bf16 / transposed / packed images of the master weights, refreshed by one launch after
    the optimiser step) against per-use casts: the same rollout, the same seeds, two PPO updates of three epochs - with a parameter
    overwritten from outside between them (the version counters must tell).  An image holds exactly what the cast would have
    produced, so the two nets differ by no more than two runs of the SAME setting do (the weight-gradient and LayerNorm backward
    kernels add their partial sums with fp32 atomics: a step is reproducible to ~1e-4 of a parameter, not to the bit)."""
    import copy
    from settlers_of_catan_rl_amd import nn_kernels
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    N, T = 4096, 16
    env = VecCatanEnv(N, seed=23); env.random_rollout(0, 700)
    net0 = CatanPolicy().cuda()
    col = RolloutCollector(env, net0, T, seed=4, autocast_dtype=torch.bfloat16)
    st = col.gather_rollouts()
    saved = nn_kernels.weight_images.enabled
    res = []
    try:
        for enabled in (True, False, False):
            nn_kernels.weight_images.enabled = enabled
            net = copy.deepcopy(net0)
            tr = PPOTrainer(net, PPOConfig(ppo_epoch=3, num_mini_batch=2), autocast_dtype=torch.bfloat16, seed=5)
            tr.update(st)
            with torch.no_grad():                                  # a change the registry did not make
                net.observation_module.tile_encoder.out_proj.weight.mul_(1.25)
                net.action_head_module.action_heads[2].mlp_2.weight.add_(0.01)
            tr.update(st)
            res.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
    finally:
        nn_kernels.weight_images.enabled = saved
    assert len(nn_kernels.weight_images.entries) > 100
    noise = float((res[1] - res[2]).abs().max())
    diff = float((res[0] - res[1]).abs().max())
    moved = float((res[1] - torch.cat([p.detach().reshape(-1) for p in net0.parameters()])).abs().max())
    assert moved > 1e-2 and diff <= 4.0 * noise + 1e-6, (diff, noise, moved)


def test_rollout_log_probs_against_the_learners_at_rollout_width(hip_lib):
    """The PPO ratio of a decision starts at exp(logp_learner - logp_rollout): the rollout's `act` (bf16 inference copy, one kernel per head,
    the observation trunk's final layer as three accumulating products above policy._PARTS_MIN_ROWS) and the learner's `evaluate_actions`
    (bf16 autocast over the fp32 masters, compact heads, one 992-wide product) are different instruction streams over the same weights.
    At 65 536 rows: the gap stays bf16 noise - pinned here so that a change of either path that widens it fails (measured on the MI355X:
    median 2.9e-4, 99.9th percentile 6.4e-3, max 9.8e-3; the learner's bf16 against its own fp32: max 1.2e-2)."""
    import torch
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    torch.manual_seed(0)
    N = 65536
    env = VecCatanEnv(N, seed=21); env.random_rollout(0, 900)
    net = CatanPolicy().cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.03 * torch.randn_like(p))            # away from the near-uniform heads of a fresh net
    f, lists, lens = env.get_obs()
    masks = env.get_action_masks()
    actor = net.inference_copy(torch.bfloat16)
    gen = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, actions, lp_roll = actor.act(f.to(torch.bfloat16), lists, lens.long(), masks, generator=gen)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, lp_learn, _ = net.evaluate_actions(f.to(torch.bfloat16), lists, lens.long(), masks, actions)
    with torch.no_grad():
        _, lp_32, _ = net.evaluate_actions(f.float(), lists, lens.long(), masks, actions)
    gap = (lp_learn.detach().float() - lp_roll.float()).abs().flatten()
    g32 = (lp_learn.detach().float() - lp_32.float()).abs().flatten()
    q = torch.quantile(gap[:1 << 16].float(), torch.tensor([0.5, 0.999], device="cuda"))
    print(f"|logp_learner - logp_rollout| at {N} rows: median {float(q[0]):.2e}, 99.9 % {float(q[1]):.2e}, max {float(gap.max()):.2e}; "
          f"learner bf16 vs fp32: max {float(g32.max()):.2e}")
    assert float(q[0]) < 2e-3 and float(q[1]) < 0.03 and float(gap.max()) < 0.1
    assert float(g32.max()) < 0.1


def test_learner_minibatch_from_gathered_parts_equals_the_row_matrix(hip_lib):
    """train.PPOTrainer._gather_obs_parts hands the observation module its pieces as the gather wrote them (policy.ObsParts: head, current
    player, the opponents' rows zero-padded to 160 columns) and the heads the PACKED mask rows (policy.PackedActionMasks): the same
    evaluate_actions as with the [B, 1 787] row matrix and the dense fp32 masks - values bit-equal (the same kernels see the same operands),
    log-probs and gradients equal up to the order of their atomic sums."""
    import torch
    from settlers_of_catan_rl_amd import nn_kernels, spec
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy, PackedActionMasks
    from settlers_of_catan_rl_amd.rollout import RolloutStorage, pack_action_masks
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    N = 65536
    env = VecCatanEnv(N, seed=2); env.random_rollout(0, 800)
    net = CatanPolicy().cuda()
    f, lists, lens = env.get_obs()
    masks = env.get_action_masks()
    fb = f.to(torch.bfloat16)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, actions, _ = net.act(fb, lists, lens.long(), masks, generator=torch.Generator(device="cuda").manual_seed(1))
    assert N >= net.action_head_module.compact_min_rows
    tr = PPOTrainer(net, PPOConfig(), autocast_dtype=torch.bfloat16, seed=0)
    idx = torch.randperm(N, device="cuda")
    o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
    tiles = fb[idx][:, o:o + 1140]
    uq, inv = torch.unique(tiles, dim=0, return_inverse=True)
    order = torch.argsort(inv, stable=True)
    start = torch.cat((torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(torch.bincount(inv, minlength=uq.shape[0]), 0)))
    dd = (uq, inv, order, start)
    packed = pack_action_masks(masks[idx]).to(torch.int32)
    st = RolloutStorage.__new__(RolloutStorage)
    res = {}
    for mode in ("matrix", "parts"):
        for p in net.parameters():
            p.grad = None
        nn_kernels.grad_arena.begin_step(fb.device); nn_kernels.wgrad_queue.begin()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if mode == "matrix":
                obs_in, m_in = fb[idx], masks[idx]
            else:
                obs_in, m_in = tr._gather_obs_parts(net, fb, idx, o), PackedActionMasks(packed, st.unpack_action_masks)
            v, lp, ent = net.evaluate_actions(obs_in, lists[idx], lens[idx].long(), m_in, actions[idx], tile_dedupe=dd)
        (lp.float().mean() + 0.5 * v.float().mean() - 0.01 * ent).backward()
        nn_kernels.wgrad_queue.flush(); nn_kernels.grad_arena.end_step()
        res[mode] = (v.detach().float().clone(), lp.detach().float().clone(), float(ent), {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None})
    assert type(tr._gather_obs_parts(net, fb, idx, o)).__name__ == "ObsParts"
    assert torch.equal(res["matrix"][0], res["parts"][0])
    # (a row's joint log-prob is the sum of up to four heads' terms, added by index_add's atomics: the order, and with it the last bit, varies)
    assert float((res["matrix"][1] - res["parts"][1]).abs().max()) <= 2e-5
    assert abs(res["matrix"][2] - res["parts"][2]) <= 1e-6 * max(1.0, abs(res["matrix"][2]))
    assert set(res["matrix"][3]) == set(res["parts"][3])
    for k, g in res["matrix"][3].items():
        scale = float(g.abs().max()) + 1e-12
        assert float((g - res["parts"][3][k]).abs().max()) <= 2e-3 * scale + 1e-9, (k, float((g - res["parts"][3][k]).abs().max()), scale)


def test_recurrent_given_kernel_equals_the_torch_formulation(hip_lib):
    """catan_recurrent_given (the inputs of a recurrent trade head evaluated for given picks) against policy._recurrent_given's torch
    formulation of RL/models/action_heads_module.py:258-329 on the CPU: conditioning columns, masks, step-major picks, step weights and
    the final counts, bit for bit, with and without `fixed` columns, from-hand and free picks, bf16 and fp32 conditioning."""
    import torch
    import torch.nn.functional as F
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator().manual_seed(4)
    B = 5000
    acts18 = torch.randint(0, 6, (B, 18), generator=g)
    acts18[::7, 8] = 0; acts18[::5, 7] = 0                                   # early stops
    cur = torch.randint(0, 4, (B, 6), generator=g).float()
    cur[::11] = 0.0                                                          # empty hands
    fixed = torch.randint(0, 3, (B, 6), generator=g).float()
    for from_hand in (True, False):
        for fx in (None, fixed):
            for col0 in (7, 11):
                acts = acts18[:, col0:col0 + 4]
                onehot = F.one_hot(acts, 6).float()
                before = torch.cumsum(onehot, 1) - onehot
                res = torch.clamp(cur[:, None, :] - before, min=0)
                mask = (res > 0).float() if from_hand else torch.ones_like(res)
                first0 = (cur.sum(-1) == 0).float()
                mask = torch.cat((torch.cat((first0[:, None, None], torch.ones(B, 3, 1)), 1), mask[:, :, 1:]), 2)
                out = torch.cat((torch.zeros(B, 4, 1), before[:, :, 1:]), 2)
                steps = lambda t: t.transpose(0, 1).reshape((4 * B,) + t.shape[2:])
                cond = steps(out) if fx is None else torch.cat((fx.repeat(4, 1), steps(out)), -1)
                keep = torch.cat((torch.ones(B, 1), (acts[:, :3] > 0).float()), 1)
                total = before[:, 3] + onehot[:, 3]
                out_final = torch.cat((torch.zeros(B, 1), total[:, 1:]), 1)
                for dt in (torch.bfloat16, torch.float32):
                    c, m, gv, kp, of = nn_kernels.recurrent_given(acts18.cuda()[:, col0:col0 + 4], cur.cuda(), None if fx is None else fx.cuda(), from_hand, dt)
                    assert c.dtype == dt and torch.equal(c.float().cpu(), cond.to(dt).float())
                    assert torch.equal(m.cpu(), steps(mask)) and torch.equal(gv.cpu(), steps(acts))
                    assert torch.equal(kp.cpu(), keep) and torch.equal(of.cpu(), out_final)


def test_categorical_on_packed_mask_bits_equals_the_float_mask_kernel(hip_lib):
    """catan_categorical_bits_fwd / _bwd (the learner's heads reading their mask as bits of the env's packed rows, through a row list, with
    up to three row segments at different bit offsets and an AND of two mask rows) against catan_categorical_fwd / _bwd on the float mask the
    bits expand to: actions, log-probs, entropies and logit gradients bit for bit (same arithmetic, same order)."""
    import torch
    from settlers_of_catan_rl_amd import nn_kernels, spec
    g = torch.Generator().manual_seed(12)
    n, W = 4000, 32
    dense = (torch.rand(n, spec.MASK_WORDS, generator=g) < 0.6).float()
    dense[::13] = 0.0                                                        # rows with an empty mask (NaN log-probs in both kernels)
    same = lambda a, b: torch.equal(a.nan_to_num(nan=123.0, posinf=1e30, neginf=-1e30), b.nan_to_num(nan=123.0, posinf=1e30, neginf=-1e30))
    bitsl = torch.zeros(n, W * 32, dtype=torch.int64)
    bitsl[:, :dense.shape[1]] = dense.long()
    words = (bitsl.view(n, W, 32) << torch.arange(32)).sum(-1)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).cuda()
    dense = dense.cuda()
    acts = torch.randint(0, 5, (n, 18), generator=g).cuda()
    for K, segs in ((13, [(None, 0, None)]), (54, [(700, 40, None), (None, 94, None)]), (5, [(900, 301, None), (300, 306, 316), (None, 306, 311)]),
                    (73, [(None, 148, None)]), (3, [(0, 280, None), (None, 283, None)])):
        B = 2500
        rows = torch.randperm(n, generator=g)[:B].cuda()
        counts, left = [], B
        for c, _, _ in segs:
            c = left if c is None else c
            counts.append(c); left -= c
        segs = [(c, o, a) for c, (_, o, a) in zip(counts, segs)]
        mask, at = [], 0
        for c, o, a in segs:
            r = rows[at:at + c]; at += c
            mk = dense[r, o:o + K]
            mask.append(mk if a is None else mk * dense[r, a:a + K])
        mask = torch.cat(mask)
        for idx in (rows, None):
            if idx is None:                                                  # no row list: row j reads packed row j
                pk, mk = words[rows].contiguous(), mask
            else:
                pk, mk = words, mask
            z = torch.randn(B, K, generator=g).cuda()
            z1, z2 = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
            given = acts[:B, 3] % K                                          # a strided column
            a1, lp1, e1 = nn_kernels.masked_categorical(z1, mk, given.contiguous(), False, None)
            a2, lp2, e2 = nn_kernels.masked_categorical_bits(z2, pk, idx, segs, given)
            assert torch.equal(a1, a2) and same(lp1, lp2) and same(e1, e2)
            ok = torch.isfinite(lp1)
            w1, w2 = torch.randn(B, generator=g).cuda(), torch.randn(B, generator=g).cuda()
            for lp, e, zz in ((lp1, e1, z1), (lp2, e2, z2)):
                (torch.where(ok, lp, torch.zeros_like(lp)) * w1 + e * w2).sum().backward()
            assert same(z1.grad, z2.grad)
