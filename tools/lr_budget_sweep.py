"""Diagnostics: tier-1 longest-road budget vs overflow rate and slow-path kernel times (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import _lib

n = 65536
window = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env = VecCatanEnv(n, seed=0)
L = _lib.lib()
env.random_rollout_deferred(4000, 16)
NP = 8
for budget in (12, 24, 48, 96, 192, 384, 768, 1536):
    env.set_lr_budgets(budget, budget)
    L.catan_profile_enable(env.h, 1)
    env.random_rollout_deferred(64, window)
    out = (C.c_uint64 * (2 * NP + 4))()
    L.catan_profile_read(env.h, out)
    L.catan_profile_enable(env.h, 0)
    kms = env.random_rollout_timed(0, 256, window)
    nslow = -(-256 // window)
    print(f"budget {budget:5d}: requests/iter {out[2*NP]/64:8.1f}  iterations/request {out[2*NP+1]/max(1,out[2*NP]):6.2f}  overflows/iter {out[2*NP+2]/64:7.2f}"
          f" | k_lr_finish {kms['k_lr_finish']*1e3/256:7.1f} us  k_lr_heavy {kms['k_lr_heavy']*1e3/nslow:7.1f} us  reset {kms['k_reset_list']*1e3/nslow:6.1f}")
