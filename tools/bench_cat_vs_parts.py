"""Diagnostics (GPU box): concatenate + one product against one accumulating product per part, at the observation module's widths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
bf = torch.bfloat16
for B in (65536, 262144, 1048576):
    for widths, O, mult in (((475, 128, 384), 512, 1), ((256, 25, 25), 128, 1), ((256, 25), 128, 3)):
        R = B * mult
        parts = [torch.randn(R, w, device="cuda").to(bf) for w in widths]
        W = torch.randn(O, sum(widths), device="cuda").to(bf); bias = torch.randn(O, device="cuda").to(bf)
        def cat(): return torch.nn.functional.linear(torch.cat(parts, -1), W, bias)
        ws, c = [], 0
        for w in widths:
            ws.append(W[:, c:c + w].t()); c += w
        def acc():
            y = torch.addmm(bias, parts[0], ws[0])
            for p, w in zip(parts[1:], ws[1:]): y.addmm_(p, w)
            return y
        wsc = [w.contiguous() for w in ws]
        def acc_c():
            y = torch.addmm(bias, parts[0], wsc[0])
            for p, w in zip(parts[1:], wsc[1:]): y.addmm_(p, w)
            return y
        print(f"rows {R:8d} widths {widths} -> {O}: cat + product {timeit(cat):8.1f} us   accumulating products {timeit(acc):8.1f} us   (contiguous W^T slices {timeit(acc_c):8.1f} us)", flush=True)
