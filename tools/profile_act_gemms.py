"""Diagnostics (GPU box): the library GEMMs of one policy `act` at rollout width (65 536 rows, bf16 inference copy, obs in bf16 as
the collector passes them): shapes, device time per call, with the shipped TunableOp file on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels
import torch.cuda.tunable as tun
B = int(os.environ.get("ROWS", "65536"))
env = VecCatanEnv(65536, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs_rows(torch.bfloat16); masks = env.get_action_masks()
f, lists, lens, masks = f[:B], lists[:B], lens[:B], masks[:B]
net = CatanPolicy().cuda(); print("tuned file loaded:", nn_kernels.use_tuned_gemms())
inf = net.inference_copy(torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(0)
def act():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return inf.act(f, lists, lens, masks, generator=g)
for on in (True, False):
    tun.enable(on)
    for _ in range(3): act()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): act()
    torch.cuda.synchronize(); print(f"tunable {'on ' if on else 'off'}: act {(time.perf_counter()-t0)/10*1e3:.2f} ms (eager)")
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        act(); torch.cuda.synchronize()
    by = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear", "aten::matmul")]
    by.sort(key=lambda e: -e.self_device_time_total)
    for e in by[:14]:
        if e.self_device_time_total > 0:
            print("   %-12s %8.1f us x%d  %s" % (e.key, e.self_device_time_total, e.count, str(e.input_shapes)[:120]))
