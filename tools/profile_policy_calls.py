"""Diagnostics (GPU box): the launches of one policy `act` at rollout width, by operator, with the Python line that issued them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda()
def act():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return net.act(f, lists, lens, masks)
for _ in range(3): act()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    act(); torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40, max_src_column_width=90))
