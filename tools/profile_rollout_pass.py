"""Diagnostics (GPU box): where one pass of RolloutCollector.gather_rollouts goes at 65 536 games (host wall time per section)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = VecCatanEnv(n, seed=0); env.random_rollout(0, 600)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, 200, seed=0, autocast_dtype=torch.bfloat16)
col.gather_rollouts(max_iters=5)
sync = torch.cuda.synchronize
T = {}
def tick(name, t0):
    sync(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
iters = 40
sync(); t_all = time.perf_counter()
col.gather_rollouts(max_iters=iters)
sync(); total = (time.perf_counter() - t_all) / iters * 1e3
# sections measured separately (each synchronised, so their sum exceeds the pipelined pass)
ar = torch.arange(n, device="cuda")
for _ in range(iters):
    t0 = time.perf_counter(); f, lists, lens = env.get_obs(); tick("get_obs", t0)
    t0 = time.perf_counter(); masks = env.get_action_masks(); dec = env.deciding_player().long(); tick("masks + deciding", t0)
    t0 = time.perf_counter(); a, lp = col._act(f, lists, lens, masks, col.policy_of_pid[ar, dec - 1], dec, col.storage.masks[0], torch.ones(n, dtype=torch.bool, device="cuda")); tick("act", t0)
    t0 = time.perf_counter(); r, d = env.step(a.to(torch.int32)); tick("env.step", t0)
print(f"pass (pipelined, incl. bookkeeping): {total:.2f} ms")
for k, v in T.items():
    print(f"  {k:18s} {v / iters * 1e3:6.2f} ms")
print(f"  bookkeeping (rest)  {total - sum(T.values()) / iters * 1e3:6.2f} ms")
