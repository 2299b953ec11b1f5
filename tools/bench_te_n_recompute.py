"""Diagnostics (GPU box): the tile encoder's training forward and its two one-pass backward kernels with the LayerNorm outputs
n1 / n2 stored by the forward and read by the passes, against recomputed in the passes (204 800 boards)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import nn_kernels, _lib

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
net = CatanPolicy().cuda()
te = net.observation_module.tile_encoder
boards = 204800
tok = boards * 19
tiles = (torch.rand(boards, 19, 60, device=dev, generator=g) < 0.1).to(torch.bfloat16)


def time_us(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for level, what in (("0", "n1 and n2 stored"), ("1", "n2 stored"), ("2", "neither stored (the default)")):
    os.environ["CATAN_TE_RECOMPUTE_N"] = level
    with torch.autocast("cuda", dtype=torch.bfloat16):
        us = time_us(lambda: nn_kernels.tile_encoder_train(te, tiles))
    print(f"training forward, {what}: {us:8.1f} us")
P, S = nn_kernels._ptr, nn_kernels._stream
L = _lib.lib()
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
dx, h, xm, n2, oo = rnd(tok, 64), torch.relu(torch.randn(tok, 128, device=dev, generator=g)).to(torch.bfloat16), rnd(tok, 64), rnd(tok, 64), rnd(tok, 64)
w2t, w1t, wot, wqt, dq = rnd(128, 64), rnd(64, 128), rnd(64, 64), rnd(64, 192), rnd(tok, 192)
lw, lb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
dxo, do = torch.empty_like(xm), torch.empty_like(oo)
acc = torch.zeros(64 * 128 + 64 + 128 * 64 + 128 + 64 * 64 + 64 + 192 * 64 + 192, device=dev)
dl = torch.zeros(2, 64, device=dev)
for name, n in (("stored", P(n2)), ("recomputed", None)):
    us = time_us(lambda: _lib.check(L.catan_ffn_outproj_bwd(P(dx), P(h), P(xm), n, P(w2t), P(w1t), P(lw), P(lb), 1e-5, P(dxo), P(acc[:8192]), P(acc[8192:8256]),
                                                            P(acc[8256:16448]), P(acc[16448:16576]), P(dl[0]), P(dl[1]), P(oo), P(wot), P(do), P(acc[16576:20672]),
                                                            P(acc[20672:20736]), tok, S())))
    print(f"k_ffn_bwd_w<out-projection>, n {name}: {us:8.1f} us")
    us = time_us(lambda: _lib.check(L.catan_qkv_bwd(P(dq), P(xm), P(dx), n, P(wqt), P(lw), P(lb), 1e-5, P(dxo), P(acc[20736:33024]), P(acc[33024:33216]), P(dl[0]), P(dl[1]),
                                                    tok, S())))
    print(f"k_qkv_bwd_w, n {name}: {us:8.1f} us")
