import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd import spec
n, T = 8192, 200
env = VecCatanEnv(n, seed=0); env.random_rollout(0, 600)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=0, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
tiles = st.obs_f[:T + 1, :, o:o + 1140]
same = (tiles[1:] == tiles[:-1]).all(-1)
print("random-init policy: identical tile features in consecutive stored observations:", float(same.float().mean()))
col.after_rollouts(); st = col.gather_rollouts()
tiles = st.obs_f[:T + 1, :, o:o + 1140]
same = (tiles[1:] == tiles[:-1]).all(-1)
print("second rollout:", float(same.float().mean()))
# how many DISTINCT boards does a random minibatch (1/64 of the rows, the reference's sampler) hold?
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
tr = PPOTrainer(net, PPOConfig(), autocast_dtype=torch.bfloat16, seed=0)
first_rows, board_of_row = tr.board_runs(st)
ids = board_of_row[: T * n]
perm = torch.randperm(T * n, device="cuda")
mb = (T * n) // 64
fr = [float(torch.unique(ids[perm[k * mb:(k + 1) * mb]]).numel()) / mb for k in range(8)]
print("distinct boards / rows in a random minibatch of 1/64:", sum(fr) / len(fr), " (distinct boards overall / rows:", first_rows.numel() / ids.numel(), ")")
