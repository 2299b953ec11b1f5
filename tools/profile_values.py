"""Diagnostics (GPU box): torch-profiler view of PPOTrainer.compute_values at config 3 (13.2 M observations per epoch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T = 65536, 200
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
tr = PPOTrainer(net, PPOConfig(), autocast_dtype=torch.bfloat16, seed=3)
tr.compute_values(st); torch.cuda.synchronize()
t0 = time.perf_counter(); tr.compute_values(st); torch.cuda.synchronize(); print("compute_values: %.1f ms" % ((time.perf_counter() - t0) * 1e3), "chunk", tr.cfg.value_chunk)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.compute_values(st); torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_type is not None and str(e.device_type).endswith("CUDA")]
ev.sort(key=lambda e: -e.self_device_time_total)
print("kernel time: %.1f ms in %d launches" % (sum(e.self_device_time_total for e in ev) / 1e3, sum(e.count for e in ev)))
for e in ev[:28]:
    print("%9.1f us x%4d  %s" % (e.self_device_time_total, e.count, e.key[:110]))
