"""Workload for counter passes over k_attn_mfma_bwd (204 800 sequences of 19 tokens, 4 heads x 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import nn_kernels
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(204800, 19, 3, 4, 16, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
o = nn_kernels.small_attention(qkv)
go = torch.randn_like(o)
for _ in range(4): torch.autograd.grad(o, qkv, go, retain_graph=True)
torch.cuda.synchronize()
