"""Diagnostics (GPU box): do TunableOp's solutions apply inside a captured hipGraph?  The player modules' first layers ([rows, 152] -> 256,
[3 rows, 159] -> 256) eager with TunableOp on / off and replayed from a graph captured with it on, plus formulations whose library
DEFAULT might be better (K padded to a multiple of 32 / 64, weight pre-transposed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import torch.cuda.tunable as tun
from settlers_of_catan_rl_amd import nn_kernels
print("tuned file loaded:", nn_kernels.use_tuned_gemms())


def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for rows, K, N in ((65536, 152, 256), (196608, 159, 256), (196608, 256, 128), (65536, 512, 1536), (65536, 987, 512), (32768, 152, 256), (98304, 159, 256)):
    x = torch.randn(rows, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    res = {}
    tun.enable(True); res["eager tuned"] = ev(lambda: F.linear(x, w, b))
    tun.enable(False); res["eager default"] = ev(lambda: F.linear(x, w, b))
    tun.enable(True)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        F.linear(x, w, b)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y = F.linear(x, w, b)
    res["graph (captured with tunable on)"] = ev(g.replay)
    tun.enable(False)
    for pad in (32, 64):
        Kp = (K + pad - 1) // pad * pad
        if Kp != K:
            xp = F.pad(x, (0, Kp - K)); wp = F.pad(w, (0, Kp - K))
            res[f"default, K padded to {Kp}"] = ev(lambda: F.linear(xp, wp, b))
    wt = w.t().contiguous()
    res["default, addmm with W^T contiguous"] = ev(lambda: torch.addmm(b, x, wt))
    tun.enable(True)
    print(f"[{rows}, {K}] -> {N}: " + "; ".join(f"{k} {v:.0f} us" for k, v in res.items()), flush=True)

print("---- column windows of the [rows, 1787] bf16 observation matrix (row pitch 3 574 B: rows are 2-byte aligned only)")
for rows, off, K, N in ((65536, 1158, 152, 256), (65536, 0, 12, 32), (65536, 12, 6, 32), (32768, 1158, 152, 256), (204800, 1158, 152, 256)):
    big = torch.randn(rows, 1787, device="cuda", dtype=torch.bfloat16)
    x = big[:, off:off + K]
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05; b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    res = {}
    tun.enable(True); res["eager tuned"] = ev(lambda: F.linear(x, w, b))
    tun.enable(False); res["eager default"] = ev(lambda: F.linear(x, w, b))
    res["default on .contiguous() (copy included)"] = ev(lambda: F.linear(x.contiguous(), w, b))
    tun.enable(True)
    g = torch.cuda.CUDAGraph(); s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        F.linear(x, w, b)
    torch.cuda.current_stream().wait_stream(s_)
    with torch.cuda.graph(g):
        y = F.linear(x, w, b)
    res["graph (tunable on)"] = ev(g.replay)
    print(f"window [{rows}, {off}:{off + K}] -> {N}: " + "; ".join(f"{k} {v:.0f} us" for k, v in res.items()), flush=True)
