"""Diagnostics (GPU box): one batch of root decisions of config 5 (4 096 roots x 64 simulations of depth 15) under the torch profiler:
wall time against device time, kernel count, device time by op - where the time goes besides the policy replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import forward_search as fs
torch.manual_seed(0)
R = 4096
root = VecCatanEnv(R, seed=0); root.random_rollout(0, 500)
net = CatanPolicy().cuda().eval()
search = fs.ForwardSearch(net, lambda n: VecCatanEnv(n, seed=1, env_id0=1 << 32, dense_reward=True, auto_reset=False), R,
                          max_depth=15, sims_per_root=64, sims_per_round=16, autocast_dtype=torch.bfloat16, use_graphs=True)
for _ in range(2):
    chosen, info = search.act(root)
    root.step(torch.from_numpy(chosen).to(root.device).to(torch.int32))
torch.cuda.synchronize(); t0 = time.perf_counter()
chosen, info = search.act(root)
torch.cuda.synchronize(); print("act (no profiler): %.3f s" % (time.perf_counter() - t0))
root.step(torch.from_numpy(chosen).to(root.device).to(torch.int32))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    chosen, info = search.act(root)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
ev = prof.key_averages()
kern = [e for e in ev if e.device_type is not None and str(e.device_type).endswith("CUDA")]
print("under the profiler: wall %.3f s, kernel time %.3f s in %d launches" % (wall, sum(e.self_device_time_total for e in kern) / 1e6, sum(e.count for e in kern)))
kern.sort(key=lambda e: -e.self_device_time_total)
for e in kern[:22]:
    print("%9.1f ms x%6d  %s" % (e.self_device_time_total / 1e3, e.count, e.key[:100]))
ops = [e for e in ev if not (e.device_type is not None and str(e.device_type).endswith("CUDA"))]
ops.sort(key=lambda e: -e.self_cpu_time_total)
print("---- host time by op")
for e in ops[:16]:
    print("%9.1f ms cpu x%6d  %s" % (e.self_cpu_time_total / 1e3, e.count, e.key[:60]))
