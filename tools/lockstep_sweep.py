"""Diagnostics (GPU box): lock-step step time over the tier-1 budget and the tier-2 round length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(8192, 32)
step = 1 << 20
for rnd in (2, 4, 8):
    for budget in (4, 6, 8, 12, 16, 24, 48):
        env.set_lr_budgets(budget, 24); env.set_lr_rounds(rnd, 48)
        env.random_rollout(step, 16); step += 16
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.random_rollout(step, 256); step += 256
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 256
        k = env.random_rollout_timed(step, 64, 0); step += 64
        print(f"round {rnd} budget {budget:3d}: {dt * 1e6:6.1f} us per step; sample {k['k_sample_random'] / 64 * 1e3:.0f} step {k['k_step'] / 64 * 1e3:.0f} lr_finish {k['k_lr_finish'] / 64 * 1e3:.0f} heavy {k['k_lr_heavy'] / 64 * 1e3:.0f} reset {k['k_reset_list'] / 64 * 1e3:.0f}", flush=True)
