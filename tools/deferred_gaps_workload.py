import os, sys
sys.path.insert(0, os.environ["REPO"])
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(3072, 32)
torch.cuda.synchronize()
env.random_rollout_deferred(1024, 32)
torch.cuda.synchronize()
