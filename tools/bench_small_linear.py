"""Micro-benchmark: the action heads' output Linear layers (128 -> 2..73, 65 536 rows, bf16) as library GEMMs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
x = torch.randn(B, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
for O in (2, 5, 13, 19, 54, 73, 128):
    w = torch.randn(O, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True); b = torch.zeros(O, device="cuda", dtype=torch.bfloat16)
    fwd = t(lambda: F.linear(x, w, b))
    y = F.linear(x, w, b); dy = torch.randn_like(y)
    dx = t(lambda: dy @ w)
    dw = t(lambda: dy.t() @ x)
    print(f"out {O:4d}: fwd {fwd:7.1f} us   dX {dx:7.1f} us   dW {dw:7.1f} us")
