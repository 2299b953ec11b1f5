import os, sys
sys.path.insert(0, '/root/repo')
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels, spec
MB = 204800; B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs()
net = CatanPolicy().cuda(); nn_kernels.use_tuned_gemms()
fm = f.repeat(4, 1)[:MB]
o = spec.OBS_FLOAT_OFFSETS
tiles = fm[:, o["tile_representations"]:o["tile_representations"] + 1140].reshape(MB, 19, 60).to(torch.bfloat16)
te = net.observation_module.tile_encoder
for _ in range(4):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = te(tiles)
    y.float().sum().backward()
torch.cuda.synchronize()
