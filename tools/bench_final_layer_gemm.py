"""Diagnostics (GPU box): the observation module's final layer (992 -> 512, bf16, bias) as the library runs it at rollout width
(65 536 rows), alone and as two concurrent products (policy and value branch), against minibatch width (204 800 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from settlers_of_catan_rl_amd import nn_kernels

dev = "cuda"
if os.environ.get("TUNED", "1") == "1":
    nn_kernels.use_tuned_gemms()


def time_us(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for rows in (65536, 204800):
    for k in (992, 987, 1024):
        x = torch.randn(rows, k, device=dev).to(torch.bfloat16); w = torch.randn(512, k, device=dev).to(torch.bfloat16); b = torch.randn(512, device=dev).to(torch.bfloat16)
        us = time_us(lambda: F.linear(x, w, b))
        print(f"rows {rows:7d} k {k:5d}: {us:7.1f} us = {2 * rows * k * 512 / us / 1e6:6.0f} TFLOP/s", flush=True)
rows, k = 65536, 992
x = torch.randn(rows, k, device=dev).to(torch.bfloat16); w = torch.randn(512, k, device=dev).to(torch.bfloat16); b = torch.randn(512, device=dev).to(torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        F.linear(x, w, b)
    with torch.cuda.stream(s2):
        F.linear(x, w, b)
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"two concurrent products at 65 536 rows: {time_us(two):7.1f} us")
def parts():
    y = torch.addmm(b, x[:, :480], w[:, :480].t())
    return y.addmm_(x[:, 480:], w[:, 480:].t())
print(f"as two accumulating products (480 + 512 columns): {time_us(parts):7.1f} us")
