import os, sys
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels, spec
MB = 204800; B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda(); nn_kernels.use_tuned_gemms()
fm, lm, nm, mm = (t.repeat((4,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks))
fm = fm.to(torch.bfloat16)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, acts, _ = net.act(fm, lm, nm, mm)
def it():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, lp, ent = net.evaluate_actions(fm, lm, nm, mm, acts)
    (v.float().sum() + lp.float().sum() + ent).backward()
for _ in range(2): it()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    it(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::fill_", "aten::cat", "aten::sum", "aten::index", "aten::add_", "aten::add", "aten::mul", "aten::_index_put_impl_", "aten::mm", "aten::addmm")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:60]:
    print("%-24s %8.1f us total  x%-4d %s" % (e.key[:24], e.self_device_time_total, e.count, str(e.input_shapes)[:120]))
