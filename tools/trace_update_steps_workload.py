"""Workload for a rocprofv3 kernel trace of real PPOTrainer minibatch steps (65 536 games, T = 50, 16 minibatches of 204 800 rows:
config 3's minibatch shape from a quarter of its rollout), STEPS optimiser steps after the warm-up ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T = 65536, 50
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16, act_buckets=(N,))
st = col.gather_rollouts()
tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=16), autocast_dtype=torch.bfloat16, seed=3)
tr.update(st)
torch.cuda.synchronize()
print("update steps workload done")
