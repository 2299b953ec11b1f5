"""Diagnostics (GPU box): the eighteen head evaluations of a policy pass as eighteen launches (catan_head_chain) and as one
(catan_head_chain_all), at the rollout's widths; HIP-event time per pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import policy as P, nn_kernels
from settlers_of_catan_rl_amd.env import VecCatanEnv
torch.set_grad_enabled(False)
net = P.CatanPolicy().cuda().inference_copy(torch.bfloat16)
ahm = net.action_head_module
for B in (65536, 16384, 4096):
    env = VecCatanEnv(B, seed=1); env.random_rollout(0, 700)
    masks = env.get_action_masks()
    f, _, _ = env.get_obs()
    cur_res, trade = f[:, 12:18].contiguous(), f[:, 0:12].contiguous()
    pre_all = (torch.randn(B, 1536, device="cuda") * 0.5).to(torch.bfloat16)
    g = torch.Generator(device="cuda").manual_seed(0)
    def run():
        return nn_kernels.heads_chain(ahm.action_heads, ahm.D, pre_all, masks, cur_res, trade, False, g)
    for one in (False, True, False, True):
        nn_kernels.HEADS_ONE_LAUNCH = one
        for _ in range(5): run()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): run()
        b.record(); torch.cuda.synchronize()
        print(f"B={B:6d} one_launch={int(one)}: {a.elapsed_time(b) / 30 * 1e3:8.1f} us per chain (incl. the torch.rand and the zeroed state)", flush=True)
    del env
