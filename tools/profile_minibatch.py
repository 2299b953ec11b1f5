"""Diagnostics (GPU box): torch-profiler table of one PPO minibatch step at the config-3 width (204 800 rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda()
nn_kernels.use_tuned_gemms()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, a, _ = net.act(f, lists, lens, masks)
rep = -(-MB // B)
fm, lm, nm, mm, am = (t.repeat((rep,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks, a))
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, lp, ent = net.evaluate_actions(fm, lm, nm, mm, am)
    loss = v.float().mean() + lp.float().mean() - 0.01 * ent
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"minibatch step, {MB} rows: {(time.perf_counter()-t0)/5*1e3:.1f} ms")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=90))
