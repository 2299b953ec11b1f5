"""`-m gpu` twin of test_reference_api_cpu.py: the L3 / L4 adapters over the HIP env and the HIP learner kernels, against the
fixtures the reference's own GamesAndPoliciesManager / BatchProcessor / EvaluationManager produced, plus the config-3 shape
(65 536 games) with sampled oracle parity."""
import ctypes as C
import types

import numpy as np
import pytest
import torch

import rollout_fixture as rf
from settlers_of_catan_rl_amd import reference_api as ra, spec

pytestmark = pytest.mark.gpu


def _hip_env(n, seed, **kw):
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    return VecCatanEnv(n, seed=seed, **kw)


def test_rollout_fixture_on_hip_collector(hip_lib):
    """SURVEY 8(c) fixture 5 on the device: HIP env + lock-step collector + BatchProcessor view == the reference manager's
    rollouts (observations, rewards accumulated over the other seats' moves, terminal masks across game ends, carry-over,
    actions, packed masks, games_complete), six rollouts in a row; state CRC of every game after every rollout."""
    envs = []
    mgr, bp, ro = rf.check_rollout_fixture(lambda n, seed: envs.append(_hip_env(n, seed)) or envs[-1])
    assert envs[0].invalid_action_count() == 0
    assert bp.obs_dict["tile_representations"].is_cuda and bp.action_masks[1].shape[0] == 3


def test_evaluation_fixture_on_hip_env(hip_lib):
    envs = []
    rf.check_eval_fixture(lambda n, seed: envs.append(_hip_env(n, seed, auto_reset=False)) or envs[-1])
    assert envs[0].invalid_action_count() == 0


class _Args(object):
    lr, eps, gamma, gae_lambda, clip_param, ppo_epoch, num_mini_batch = 3e-4, 1e-5, 0.999, 0.95, 0.2, 2, 4
    value_loss_coef, entropy_coef_start, max_grad_norm, recompute_returns = 1.0, 0.04, 0.5, True
    num_processes, num_envs_per_process, num_steps = 64, 4, 8


def test_run_update_sequence_on_device(hip_lib):
    """robust_train.py:101-108 through the adapters with the real nets (bf16 autocast) and the HIP GAE / loss kernels; the
    advantages of `compute_advantages_alt` are checked against the reference formulas (process_batch.py:134-142) in torch."""
    args = _Args()
    torch.manual_seed(0)
    mgr = ra.SubProcGameManager([ra.make_game_manager(args.num_envs_per_process, args.num_steps) for _ in range(args.num_processes)], seed=3)
    mgr.env.random_rollout(0, 700)
    mgr.collector.reset()
    central = ra.build_agent_model(device="cuda")
    storage = ra.BatchProcessor(args, central.lstm_size, device="cuda")
    agent = ra.PPO(central, args)
    mgr.update_policy(central.state_dict(), policy_id=0)
    for u in range(2):
        storage.process_rollouts(mgr.gather_rollouts())
        vl, al, el = agent.update(storage)
        assert all(np.isfinite(x) for x in (vl, al, el)), (vl, al, el)
        mgr.update_policy(central.state_dict(), policy_id=0)
        ra.update_opponent_policies([central.state_dict()], mgr, args, rng=np.random.RandomState(u))
    T, N = args.num_steps, 256
    assert storage.values.shape == (T + 1, N, 1) and storage.advantages.shape == (T, N, 1)
    r, v, m = storage.rewards[..., 0].double(), storage.values[..., 0].double(), storage.masks[..., 0].double()
    gae, ret = 0, torch.zeros_like(r)
    for t in reversed(range(T)):
        delta = r[t] + args.gamma * v[t + 1] * m[t + 1] - v[t]
        gae = delta + args.gamma * args.gae_lambda * m[t + 1] * gae
        ret[t] = gae + v[t]
    adv = ret - v[:-1]
    adv = (adv - adv.mean()) / (adv.std() + 1e-5)
    assert torch.allclose(storage.returns[..., 0].double(), ret, rtol=1e-5, atol=1e-3)
    assert torch.allclose(storage.advantages[..., 0].double(), adv, rtol=1e-4, atol=1e-4)
    assert mgr.env.invalid_action_count() == 0
    assert len(mgr.collector.opponent_nets) == 1            # every group drew the same league entry (one dict object) -> one net in play


def test_config3_shape_65536_games_sampled_parity(oracle, hip_lib):
    """BASELINE.json configs[2] at full width: 65 536 games, 4-seat self-play with the RL/models net in bf16, T = 3 (reduced,
    stated), one PPO epoch of 64 minibatches.  Properties over all games + bit-exact parity of sampled games: the recorded
    decisions of ALL seats are replayed on the CPU oracle and the stored active-seat observations, action masks, terminal
    masks and rewards must be the oracle's."""
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    n, T, seed = 65536, 3, 21
    env = _hip_env(n, seed)
    sample = [0, 1, 777, 4095, 30000, 65535]
    rec = _Recorder(env, sample)
    torch.manual_seed(1)
    net = CatanPolicy().cuda()
    col = RolloutCollector(rec, net, T, seed=5, autocast_dtype=torch.bfloat16)
    st = col.gather_rollouts()
    assert env.invalid_action_count() == 0
    assert bool(((st.masks[:T + 1] == 0) | (st.masks[:T + 1] == 1)).all()) and bool((col.n_obs == T + 1).all()) and bool((col.n_act == T).all())
    assert bool(torch.isfinite(st.action_log_probs).all()) and bool((st.action_log_probs <= 0).all())
    am = st.unpack_action_masks(st.action_masks[:, sample])
    assert bool((am[..., :13].sum(-1) >= 1).all())                    # at least one legal action type in every stored mask
    # sampled parity against the oracle
    active = col.active_pid.cpu().numpy()
    for j, gidx in enumerate(sample):
        o = oracle.OracleEnv(seed, gidx); o.reset()
        t_obs = 0
        rsum = np.zeros(4)
        for a in rec.trace[j] + [None]:
            if o.deciding_player() == active[gidx] and t_obs <= T:
                f, lists, lens, _ = o.obs()
                assert st.obs_f.dtype == torch.bfloat16                 # stored in the autocast dtype: exact (multiples of 1/8)
                assert np.array_equal(st.obs_f[t_obs, gidx].float().cpu().numpy(), f), (gidx, t_obs)
                assert np.array_equal(st.lens[t_obs, gidx].cpu().numpy(), lens)
                if t_obs < T:
                    assert np.array_equal(st.unpack_action_masks(st.action_masks[t_obs, gidx]).cpu().numpy(), o.masks()), (gidx, t_obs)
                if t_obs > 0:
                    assert abs(float(st.rewards[t_obs - 1, gidx]) - rsum[active[gidx] - 1]) < 1e-6
                    rsum[:] = 0
                t_obs += 1
            if a is None:
                break
            assert o.is_legal(a)
            rew, done = o.step(a)
            rsum += rew
            assert not done
        assert t_obs == T + 1, (gidx, t_obs)
    # one PPO epoch at this width: 64 minibatches of 3 072 rows
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=1), autocast_dtype=torch.bfloat16, seed=0)
    before = [p.detach().clone() for p in net.parameters()]
    vl, al, el = tr.update(st)
    assert all(np.isfinite(x) for x in (vl, al, el))
    assert any(not torch.equal(b, p.detach()) for b, p in zip(before, net.parameters()))


class _Recorder(object):
    """delegates to the env and keeps the real (non no-op) actions of a few sampled games"""

    def __init__(self, env, sample):
        self.env, self.n, self.device = env, env.n, env.device
        self.idx = torch.tensor(sample, device=env.device)
        self.trace = [[] for _ in sample]

    def step(self, actions):
        a = actions[self.idx].cpu().numpy()
        for j in range(len(self.trace)):
            if a[j, 0] >= 0:
                self.trace[j].append(a[j].astype(np.int32))
        return self.env.step(actions)

    def step_deferred(self, actions, window=32, status_out=None):
        """(the collector's default on the device) a waiting game ignores the action it is passed: not part of its trace"""
        a = actions[self.idx].cpu().numpy()
        w = getattr(self, "_waiting", None)
        for j in range(len(self.trace)):
            if a[j, 0] >= 0 and not (w is not None and w[j]):
                self.trace[j].append(a[j].astype(np.int32))
        r, d, s = self.env.step_deferred(actions, window, status_out=status_out)
        self._waiting = (s[self.idx] == 1).cpu().numpy()
        return r, d, s

    def step_flush(self):
        self._waiting = None
        return self.env.step_flush()

    def __getattr__(self, name):
        return getattr(self.env, name)
