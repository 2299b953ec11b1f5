"""Diagnostics (GPU box): k_tile_encoder_fwd alone - the inference kernel at the widths the passes use and the training forward at
the minibatch's board count, with a checksum of the outputs (the kernel's results do not depend on how boards are grouped into
workgroups: two builds must print the same checksums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import policy as P, nn_kernels

torch.manual_seed(0)
net = P.CatanPolicy().cuda()
te = net.observation_module.tile_encoder


def time_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


with torch.no_grad():
    for boards in (204800, 65536, 16384, 4096, 1003):
        g = torch.Generator(device="cuda").manual_seed(boards)
        tiles = (torch.rand((boards, 19, 60), device="cuda", generator=g) < 0.2).to(torch.bfloat16)
        out = nn_kernels.tile_encoder_forward(te, tiles)
        cs = int(out.view(torch.int16).to(torch.int64).sum().item())
        us = time_us(lambda: nn_kernels.tile_encoder_forward(te, tiles))
        print(f"inference {boards:7d} boards: {us:8.1f} us  ({us * 1e3 / boards:6.2f} ns per board)  checksum {cs}", flush=True)
tiles = (torch.rand((204800, 19, 60), device="cuda") < 0.2).to(torch.bfloat16)
if hasattr(nn_kernels, "tile_encoder_train"):
    us = time_us(lambda: nn_kernels.tile_encoder_train(te, tiles), reps=5)
    print(f"training forward 204800 boards: {us:8.1f} us", flush=True)
