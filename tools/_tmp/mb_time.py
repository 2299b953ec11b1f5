import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
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
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): it()
torch.cuda.synchronize(); print("minibatch fwd+bwd (bf16 obs): %.2f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
