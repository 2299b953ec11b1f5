"""Diagnostics (GPU box): catan_linear_wgrad at the shapes of a config-3 minibatch step, against the HBM time of reading x and dy once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import nn_kernels
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [(152, 256, 204800), (159, 256, 614400), (160, 256, 614400), (256, 128, 204800), (256, 128, 614400), (25, 128, 204800), (25, 128, 614400),
          (64, 192, 3412000), (64, 64, 3412000), (64, 128, 3412000), (128, 64, 3412000), (64, 32, 3412000), (512, 128, 204800), (128, 128, 204800), (128, 128, 37000),
          (16, 25, 614400), (475, 512, 204800), (512, 256, 204800)]
for (I, O, rows) in shapes:
    if not nn_kernels.wgrad_supported(rows, I, O):
        print(f"I={I:4d} O={O:4d} rows={rows:8d}: unsupported"); continue
    x = torch.randn(rows, I, device="cuda").to(torch.bfloat16); dy = torch.randn(rows, O, device="cuda").to(torch.bfloat16)
    tk = timeit(lambda: nn_kernels.wgrad(x, dy))
    tl = timeit(lambda: (dy.t() @ x, dy.float().sum(0)))
    hbm = rows * (I + O) * 2 / 5.0e12 * 1e6
    print(f"I={I:4d} O={O:4d} rows={rows:8d}: k_wgrad {tk:7.1f} us   library mm + bias sum {tl:7.1f} us   x + dy once at 5 TB/s {hbm:6.1f} us   {2 * rows * I * O / tk / 1e6:6.1f} TFLOP/s", flush=True)
