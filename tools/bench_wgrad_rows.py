"""Diagnostics (GPU box): k_wgrad against the library product for the heads' layer shapes over the row count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import nn_kernels
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (I, O) in ((128, 128), (128, 16), (512, 128), (256, 128)):
    for rows in (4096, 16384, 37804, 65536, 131072, 204800, 614400):
        x = torch.randn(rows, I, device="cuda").to(torch.bfloat16); dy = torch.randn(rows, O, device="cuda").to(torch.bfloat16)
        tk = timeit(lambda: nn_kernels.wgrad(x, dy))
        tl = timeit(lambda: (dy.t() @ x, dy.float().sum(0)))
        print(f"I={I:4d} O={O:4d} rows={rows:7d}: k_wgrad {tk:7.1f} us   library mm + bias sum {tl:7.1f} us", flush=True)
