"""Diagnostics (GPU box): PPOTrainer.compute_values at config 3 (13.2 M observations) over the rows per forward (PPOConfig.value_chunk)."""
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
for ch in (65536, 131072, 204800, 262144, 524288, 1048576):
    tr = PPOTrainer(net, PPOConfig(value_chunk=ch), autocast_dtype=torch.bfloat16, seed=3)
    tr.compute_values(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr.compute_values(st)
    torch.cuda.synchronize(); print("chunk %8d: %.1f ms" % (ch, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
