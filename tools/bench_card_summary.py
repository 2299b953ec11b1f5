"""Diagnostics (GPU box): the fused card-list kernel alone (forward, backward) at 204 800 lists."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import nn_kernels
B = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
g = torch.Generator(device="cuda").manual_seed(1)
lens = torch.randint(1, 12, (B,), device="cuda", generator=g)
deck = torch.tensor([1] * 14 + [2] * 5 + [3] * 2 + [4] * 2 + [5] * 2, device="cuda")          # game/game.py:77
ids = deck[torch.rand(B, 25, device="cuda", generator=g).argsort(1)]                           # a shuffled deck per list ...
ids = (ids * (torch.arange(25, device="cuda")[None] < lens[:, None])).to(torch.int8)          # ... of which the first `len` cards count
params = torch.randn(544, device="cuda", requires_grad=True)
l32 = lens.to(torch.int32)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = nn_kernels._CardSummary.apply(ids, l32, params, 1e-5)
print("fwd  %.3f ms" % t(lambda: nn_kernels._CardSummary.apply(ids, l32, params, 1e-5)))
d = torch.randn_like(out)
def fb():
    params.grad = None
    nn_kernels._CardSummary.apply(ids, l32, params, 1e-5).backward(d)
print("fwd+bwd %.3f ms" % t(fb))
