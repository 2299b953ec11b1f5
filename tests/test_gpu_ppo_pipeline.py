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


@pytest.mark.parametrize("D", [16, 25, 64])
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
        assert torch.allclose(gx, xr.grad, atol=tol * 4, rtol=tol * 4)
        n = x.numel() // D
        assert torch.allclose(gw, ln.weight.grad, atol=tol * 4 * max(1, n ** 0.5), rtol=2e-2)
        assert torch.allclose(gb, ln.bias.grad, atol=tol * 4 * max(1, n ** 0.5), rtol=2e-2)
