"""Diagnostics (GPU box): torch-profiler view of real PPOTrainer minibatch steps at config 3 (65 536 games x T = 200):
wall time per step, device time per step and the ops outside the net's forward / backward (gathers, mask unpacking,
loss, gradient clipping, Adam)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 200
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=64), autocast_dtype=torch.bfloat16, seed=3)
class Stop(Exception): pass
calls = [0]
orig = tr.optimiser.step
prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(os.environ.get('STACKS')),
               experimental_config=(torch._C._profiler._ExperimentalConfig(verbose=True) if os.environ.get('STACKS') else None))
NS = 4
def step(*a, **k):
    r = orig(*a, **k)
    calls[0] += 1
    if calls[0] == 4:
        torch.cuda.synchronize(); prof.__enter__(); step.t0 = time.perf_counter()
    if calls[0] == 4 + NS:
        torch.cuda.synchronize(); step.t1 = time.perf_counter(); prof.__exit__(None, None, None); raise Stop()
    return r
tr.optimiser.step = step
try:
    tr.update(st)
except Stop:
    pass
print("ms per minibatch step under the profiler: %.2f" % ((step.t1 - step.t0) / NS * 1e3))
ev = prof.key_averages()
kern = [e for e in ev if e.device_type is not None and str(e.device_type).endswith("CUDA")]
print("kernel time per step: %.2f ms in %.0f launches" % (sum(e.self_device_time_total for e in kern) / NS / 1e3, sum(e.count for e in kern) / NS))
ops = [e for e in ev if not (e.device_type is not None and str(e.device_type).endswith("CUDA"))]
ops.sort(key=lambda e: -e.self_device_time_total)
for e in ops[:40]:
    print("%-40s dev %8.1f us/step  cpu %8.1f us/step  x%.0f" % (e.key[:40], e.self_device_time_total / NS, e.self_cpu_time_total / NS, e.count / NS))

print("---- copies / fills / gathers by shape (device us per step)")
by = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::fill_", "aten::index", "aten::cat", "aten::add_", "aten::add", "aten::mm", "aten::addmm")]
by.sort(key=lambda e: -e.self_device_time_total)
for k in ("aten::copy_", "aten::fill_", "aten::index", "aten::cat", "aten::add_", "aten::add", "aten::mm", "aten::addmm"):
    es = [e for e in by if e.key == k]
    small = [e for e in es if e.self_device_time_total / max(1, e.count) < 8.0]
    print("%-12s total %8.1f us/step in %5.1f calls; of which launches under 8 us: %8.1f us/step in %5.1f calls" % (
        k, sum(e.self_device_time_total for e in es) / NS, sum(e.count for e in es) / NS, sum(e.self_device_time_total for e in small) / NS, sum(e.count for e in small) / NS))
for e in by[:60]:
    print("%-14s %8.1f us/step x%.0f  %s" % (e.key, e.self_device_time_total / NS, e.count / NS, str(e.input_shapes)[:150]))

print("---- the net's own autograd functions by shape (device us per step)")
by = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("_") and not e.key.startswith("_foreach")]
by.sort(key=lambda e: -e.self_device_time_total)
for e in by[:70]:
    print("%-28s %8.1f us/step x%.1f  %s" % (e.key[:28], e.self_device_time_total / NS, e.count / NS, str(e.input_shapes)[:170]))

if os.environ.get("STACKS"):
    print("---- small launches (copy_ / fill_ / add_ / add / mul / zero_) by the first frame inside the package (device us per step, calls per step)")
    agg = {}
    for e in prof.key_averages(group_by_stack_n=12):
        if e.key not in ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::mul", "aten::zero_", "aten::cat", "aten::sum", "aten::index"):
            continue
        fr = next((f for f in e.stack if "settlers_of_catan_rl_amd" in f), e.stack[0] if e.stack else "?")
        fr = fr.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/settlers_of_catan_rl_amd/", "")
        k = (e.key, fr[:110])
        a = agg.setdefault(k, [0.0, 0])
        a[0] += e.self_device_time_total; a[1] += e.count
    for (key, fr), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:90]:
        print("%-12s %8.1f us/step x%6.1f  %s" % (key, t / NS, c / NS, fr))
