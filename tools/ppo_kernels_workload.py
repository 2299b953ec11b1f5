"""Workload for the rocprofv3 kernel-trace of the two fused learner kernels at config-3 sizes: k_gae (+ statistics,
normalisation) on T = 200 x N = 65 536 and k_ppo_loss on a 204 800-row minibatch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import ppo
T, N, B = 200, 65536, 204800
g = torch.Generator(device="cuda").manual_seed(0)
r = torch.where(torch.rand(T, N, device="cuda", generator=g) < 0.002, 500.0, 0.0)
v = 150 + 40 * torch.randn(T + 1, N, device="cuda", generator=g)
m = (torch.rand(T + 1, N, device="cuda", generator=g) > 0.002).float()
lp = (-3 + 0.5 * torch.randn(B, device="cuda", generator=g)).requires_grad_(True)
val = torch.randn(B, device="cuda", generator=g).requires_grad_(True)
old, adv, vold, ret = (torch.randn(B, device="cuda", generator=g) for _ in range(4))
for _ in range(int(os.environ.get("REPS", "20"))):
    ppo.compute_gae(r, v, m, 0.999, 0.95)
    loss, _ = ppo.ppo_loss(lp, val, old - 3, adv, vold * 40 + 150, ret * 40 + 150, 0.2, 1.0, value_normaliser=(150.0, 150.0))
    loss.backward()
torch.cuda.synchronize()
print("done")
