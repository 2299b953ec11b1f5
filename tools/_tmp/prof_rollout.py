import sys, time
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
N, T = 65536, 200
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
col.gather_rollouts(max_iters=30)
torch.cuda.synchronize(); t0 = time.perf_counter()
col.gather_rollouts(max_iters=100)
torch.cuda.synchronize(); print("ms per env iteration: %.2f" % ((time.perf_counter() - t0) / 100 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    col.gather_rollouts(max_iters=20); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=60))
