import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(3000, 16)
for budget in (12, 24, 48, 96):
    for w in (8, 16, 32):
        env.set_lr_budgets(24, budget)
        env.random_rollout_deferred(64, w)
        c0 = int(env.policy_counters().sum()); torch.cuda.synchronize(); t0 = time.perf_counter()
        env.random_rollout_deferred(1024, w)
        c1 = int(env.policy_counters().sum()); dt = time.perf_counter() - t0
        kms = env.random_rollout_timed(0, 256, w)
        ns = -(-256 // w)
        print(f"budget {budget:3d} window {w:2d}: {(c1-c0)/dt/1e6:7.1f} M/s  {dt/1024*1e6:6.1f} us/iter  active {(c1-c0)/1024/65536:.3f} | lr_finish {kms['k_lr_finish']*1e3/256:6.1f} heavy {kms['k_lr_heavy']*1e3/ns:7.1f} finish {kms['k_step_finish']*1e3/ns:6.1f} reset {kms['k_reset_list']*1e3/ns:6.1f}")
