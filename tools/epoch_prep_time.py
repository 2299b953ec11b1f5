"""Diagnostics (GPU box): the per-epoch preparation of PPOTrainer.update at config 3 (permutation, distinct boards and the heads' row
sets of the 64 minibatches), each timed alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T = 65536, 200
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=64), autocast_dtype=torch.bfloat16, seed=3)
total, mbs = T * N, T * N // 64
def timed(name, fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:8.1f} ms", flush=True)
    return r
first_rows, board_of_row = timed("board_runs (once per rollout)", lambda: tr.board_runs(st), 1)
perm = timed("randperm", lambda: torch.randperm(total, device="cuda"))
timed("minibatch_boards", lambda: tr.minibatch_boards(board_of_row[:total], perm, 64, mbs))
acts_all = st.actions[:T].reshape(total, -1)
timed("precompute_groupings", lambda: net.action_head_module.precompute_groupings(acts_all, perm, 64, mbs))
timed("compute_values", lambda: tr.compute_values(st), 2)
