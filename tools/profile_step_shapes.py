"""Diagnostics (GPU box): the matrix products of one minibatch training step (204 800 rows, bf16 autocast) with their input shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels
B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda(); nn_kernels.use_tuned_gemms()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, a, _ = net.act(f, lists, lens, masks)
MB = 204800
rep = -(-MB // B)
f, lists, lens, masks, a = (t.repeat((rep,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks, a))
f = f.to(torch.bfloat16)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, lp, ent = net.evaluate_actions(f, lists, lens, masks, a)
    loss = v.float().mean() + lp.float().mean() - 0.01 * ent
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear", "aten::matmul")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    print(f"{e.device_time_total / 1e3:8.2f} ms x{e.count:3d}  {e.key:12s} {e.input_shapes}")
