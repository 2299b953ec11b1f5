import sys, time
sys.path.insert(0, "/root/repo")
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(3000, 32); torch.cuda.synchronize()
for it in (1024, 4096):
    t0 = time.perf_counter(); env.random_rollout_deferred(it, 32); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iters {it}: host enqueue {1e6*(t1-t0)/it:.1f} us/iter, total {1e6*(t2-t0)/it:.1f} us/iter")
t0 = time.perf_counter(); env.random_rollout(0, 1024); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"lock-step: host enqueue {1e6*(t1-t0)/1024:.1f} us/iter, total {1e6*(t2-t0)/1024:.1f} us/iter")
