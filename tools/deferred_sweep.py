"""Diagnostics (GPU box): deferred-loop throughput over the tier-1 iteration budget, the window length W and the tier-2
round length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(8192, 32)
for rnd in (16, 48):
    for budget in (6, 8, 12, 16, 24):
        for w in (16, 32, 48):
            env.set_lr_budgets(16, budget); env.set_lr_rounds(4, rnd)
            env.random_rollout_deferred(2 * w, w)
            c0 = int(env.policy_counters().sum()); torch.cuda.synchronize(); t0 = time.perf_counter()
            env.random_rollout_deferred(4096, w)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            c1 = int(env.policy_counters().sum())
            print(f"round {rnd:2d} budget {budget:3d} window {w:2d}: {(c1-c0)/dt/1e6:7.1f} M/s  {dt/4096*1e6:6.1f} us/iter  active {(c1-c0)/4096/65536:.3f}", flush=True)
