"""Diagnostics (GPU box): torch-profiler table of the action heads' evaluate pass (forward + backward) at 204 800 rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels
MB, B = 204800, 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda(); nn_kernels.use_tuned_gemms()
rep = -(-MB // B)
fm, lm, nm, mm = (t.repeat((rep,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks))
main = torch.randn(MB, 512, device="cuda", dtype=torch.bfloat16, requires_grad=True)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, acts, _ = net.act(fm, lm, nm, mm)
cur_res, trade = net._custom(fm)
def heads():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, lp, ent = net.action_head_module(main, mm.float(), cur_res, trade, acts)
    (lp.sum() + ent).backward()
for _ in range(3): heads()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): heads()
torch.cuda.synchronize(); print(f"heads fwd+bwd: {(time.perf_counter()-t0)/5*1e3:.1f} ms")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    heads(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=80))
