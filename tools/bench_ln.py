"""Diagnostics (GPU box): the LayerNorm kernels at the player modules' / heads' shapes of a config-3 minibatch step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import nn_kernels
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rows, D in ((204800, 25), (614400, 25), (3412000, 25), (204800, 128), (614400, 128), (204800, 256), (614400, 256), (204800, 512), (65536, 128), (65536, 256), (196608, 128), (196608, 256)):
    ln = torch.nn.LayerNorm(D).cuda()
    x = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
    xg = x.clone().requires_grad_(True)
    y = nn_kernels.small_layer_norm(xg, ln, True)
    gy = torch.randn_like(y)
    tf = timeit(lambda: nn_kernels.small_layer_norm(x, ln, True))
    tb = timeit(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
    print(f"rows {rows:7d} D {D:4d}: forward {tf:7.1f} us ({rows * D * 4 / tf / 1e6:5.2f} TB/s)   backward {tb:7.1f} us ({rows * D * 6 / tb / 1e6:5.2f} TB/s)", flush=True)
