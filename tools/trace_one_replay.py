import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import policy as P, nn_kernels, spec
from settlers_of_catan_rl_amd.forward_search import GraphedAct
B = 65536
torch.set_grad_enabled(False)          # as the collector runs it (RolloutCollector.gather_rollouts is @torch.no_grad)
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 800)
f, lists, lens = env.get_obs_rows(torch.bfloat16); masks = env.get_action_masks(); lens = lens.long()
net = P.CatanPolicy().cuda().inference_copy(torch.bfloat16)
nn_kernels.use_tuned_gemms()
gen = torch.Generator(device="cuda").manual_seed(1)
ga = GraphedAct(net, buckets=(B,), autocast_dtype=torch.bfloat16, generator=gen)
ga(f, lists, lens, masks)
st = ga.graphs[B]
for _ in range(3): st["g"].replay()
torch.cuda.synchronize()
env.random_rollout(10000, 1)      # marker kernels (k_step) before the replay under study
torch.cuda.synchronize()
st["g"].replay()
torch.cuda.synchronize()
env.random_rollout(10001, 1)
torch.cuda.synchronize()
