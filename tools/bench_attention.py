"""Diagnostics (GPU box): k_attn_mfma_fwd / _bwd alone at a minibatch's 204 800 tile sequences (19 tokens, 4 heads x 16)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from settlers_of_catan_rl_amd import nn_kernels
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(204800, 19, 3, 4, 16, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
o = nn_kernels.small_attention(qkv); go = torch.randn_like(o)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print("k_attn_mfma_fwd: %.1f us   k_attn_mfma_bwd: %.1f us" % (t(lambda: nn_kernels.small_attention(qkv.detach())), t(lambda: torch.autograd.grad(o, qkv, go, retain_graph=True))))
