import sys, time
sys.path.insert(0, '/root/repo')
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
tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=64), autocast_dtype=torch.bfloat16, seed=3)
class Stop(Exception): pass
calls = [0]
orig = tr.optimiser.step
prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
def step(*a, **k):
    r = orig(*a, **k)
    calls[0] += 1
    if calls[0] == 4:
        torch.cuda.synchronize(); prof.__enter__(); step.t0 = time.perf_counter()
    if calls[0] == 8:
        torch.cuda.synchronize(); step.t1 = time.perf_counter(); prof.__exit__(None, None, None); raise Stop()
    return r
tr.optimiser.step = step
try:
    tr.update(st)
except Stop:
    pass
print("ms per minibatch step (4 steps): %.2f" % ((step.t1 - step.t0) / 4 * 1e3))
ev = prof.key_averages()
tot_cuda = sum(e.self_device_time_total for e in ev)
print("device ms per step: %.2f" % (tot_cuda / 4 / 1e3))
rows = [e for e in ev if not e.key.startswith(("void ", "Cijk", "catan::", "Custom_", "Memcpy", "Memset"))]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:45]:
    print("%-40s dev %8.1f us/step  cpu %8.1f us/step  x%.0f" % (e.key[:40], e.self_device_time_total / 4, e.self_cpu_time_total / 4, e.count / 4))
