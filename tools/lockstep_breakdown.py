"""Diagnostics (GPU box): per-kernel time of the lock-step step (catan_random_rollout) at 65 536 games."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout(0, 3000)
torch.cuda.synchronize(); t0 = time.perf_counter()
env.random_rollout(3000, 512)
torch.cuda.synchronize(); print(f"lock-step: {(time.perf_counter() - t0) / 512 * 1e6:.1f} us per step")
k = env.random_rollout_timed(3512, 512, 0)
for name, ms in k.items():
    print(f"  {name:18s} {ms / 512 * 1e3:7.1f} us per step")
print(f"  sum                {sum(k.values()) / 512 * 1e3:7.1f} us")
