"""Workload for rocprofv3 --pmc passes over the learner-side kernels (tools/profile_round6.sh): each kernel of interest launched a few
times at the shapes of a config-3 minibatch step, plus k_calib_copy launches with exactly known HBM traffic (256 MiB read + 256 MiB
written each).  tools/pmc_summarise.py (PMC_TAIL=4) averages the last launches of every kernel; the algorithmic bytes to set the
measured traffic against are those of tools/learner_rooflines.py (same shapes)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import _lib, nn_kernels, ppo as K, spec

CALIB_BYTES = 256 << 20
L = _lib.lib()
P, S = nn_kernels._ptr, nn_kernels._stream
a = torch.empty(CALIB_BYTES, dtype=torch.uint8, device="cuda").random_(0, 255)
b = torch.empty_like(a)
for _ in range(4):
    _lib.check(L.catan_calib_copy(P(b), P(a), CALIB_BYTES, S()))
torch.cuda.synchronize()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
n, T, rows_mb = 65536, 200, 204800
env = VecCatanEnv(n, seed=0); env.random_rollout(0, 800)
net = CatanPolicy().cuda()
REPS = 4
# k_obs_rows, k_gae
dense = env.get_obs_rows(torch.bfloat16)
for _ in range(REPS): env.get_obs_rows(torch.bfloat16, out=dense)
r = torch.randn(T, n, device=dev, generator=g); v = torch.randn(T + 1, n, device=dev, generator=g); m = (torch.rand(T + 1, n, device=dev, generator=g) > 0.02).float()
for _ in range(REPS): K.compute_gae(r, v, m, 0.999, 0.95, process_group=False)
del r, v, m
# the tile encoder's training forward and its one-pass backward kernels
te = net.observation_module.tile_encoder
tiles = (torch.rand(rows_mb, 19, 60, device=dev, generator=g) < 0.1).to(torch.bfloat16)
with torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(REPS): nn_kernels.tile_encoder_train(te, tiles)
del tiles
tok = rows_mb * 19
dx = torch.randn(tok, 64, device=dev, generator=g).to(torch.bfloat16); h = torch.relu(torch.randn(tok, 128, device=dev, generator=g)).to(torch.bfloat16)
xm = torch.randn(tok, 64, device=dev, generator=g).to(torch.bfloat16)
w2t = torch.randn(128, 64, device=dev, generator=g).to(torch.bfloat16); w1t = torch.randn(64, 128, device=dev, generator=g).to(torch.bfloat16)
lw = torch.ones(64, device=dev); dh = torch.empty_like(h); dxo = torch.empty_like(xm); dl = torch.zeros(2, 64, device=dev)
for _ in range(REPS): _lib.check(L.catan_ffn_bwd_dx(P(dx), P(h), P(xm), P(w2t), P(w1t), P(lw), 1e-5, P(dh), P(dxo), P(dl[0]), P(dl[1]), tok, S()))
dq = torch.randn(tok, 192, device=dev, generator=g).to(torch.bfloat16); wqt = torch.randn(64, 192, device=dev, generator=g).to(torch.bfloat16)
for _ in range(REPS): _lib.check(L.catan_qkv_bwd_dx(P(dq), P(xm), P(dx), P(wqt), P(lw), 1e-5, P(dxo), P(dl[0]), P(dl[1]), tok, S()))
# the same chains with the sub-layers' weight gradients (and the out-projection's backward) in the pass: what the update's backward runs
lb = torch.randn(64, device=dev, generator=g); n2 = torch.randn(tok, 64, device=dev, generator=g).to(torch.bfloat16); oo = torch.randn(tok, 64, device=dev, generator=g).to(torch.bfloat16)
wot = torch.randn(64, 64, device=dev, generator=g).to(torch.bfloat16); do = torch.empty_like(oo)
acc = torch.zeros(64 * 128 + 64 + 128 * 64 + 128 + 64 * 64 + 64 + 192 * 64 + 192, device=dev)
for _ in range(REPS):
    _lib.check(L.catan_ffn_outproj_bwd(P(dx), P(h), P(xm), None, P(w2t), P(w1t), P(lw), P(lb), 1e-5, P(dxo), P(acc[:8192]), P(acc[8192:8256]), P(acc[8256:16448]),
                                       P(acc[16448:16576]), P(dl[0]), P(dl[1]), P(oo), P(wot), P(do), P(acc[16576:20672]), P(acc[20672:20736]), tok, S()))
for _ in range(REPS):
    _lib.check(L.catan_qkv_bwd(P(dq), P(xm), P(dx), None, P(wqt), P(lw), P(lb), 1e-5, P(dxo), P(acc[20736:33024]), P(acc[33024:33216]), P(dl[0]), P(dl[1]), tok, S()))
del n2, oo, do
# a weight gradient and a row product at the encoder's shapes
for _ in range(REPS): nn_kernels.wgrad(h, dx)
w = torch.randn(128, 64, device=dev, generator=g).to(torch.bfloat16); bb = torch.zeros(128, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(REPS): nn_kernels.linear_inference(xm, w, bb)
del dx, h, xm, dh, dxo, dq
# row movement
rows_all = 16 * rows_mb
store = torch.randn(rows_all // 16, 16 * 1787, device=dev, generator=g).to(torch.bfloat16).view(rows_all, 1787)
idx = torch.randint(0, rows_all, (rows_mb,), device=dev, generator=g)
for _ in range(REPS): nn_kernels.gather_rows(store[:, 18:1158], idx)
del store
U = int(0.875 * rows_mb)
inv = torch.randint(0, U, (rows_mb,), device=dev, generator=g); inv[:U] = torch.arange(U, device=dev)
order = torch.argsort(inv, stable=True)
start = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(torch.bincount(inv, minlength=U), 0)))
srcu = torch.randn(U, 480, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
dy = None
for _ in range(REPS):
    y = nn_kernels.expand_rows(srcu, inv, order, start)
    dy = torch.randn_like(y) if dy is None else dy
    torch.autograd.grad(y, srcu, dy)
# the attention kernels (19 x 19, 4 heads x 16: 7 296 B of qkv in, 2 432 B out per sequence; backward: + dout in, dqkv out) and the
# inference tile encoder (2 280 B in, 950 B out per board)
del srcu, y, dy
qkv = torch.randn(rows_mb, 19, 3, 4, 16, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
go = None
for _ in range(REPS):
    o = nn_kernels.small_attention(qkv)
    go = torch.randn_like(o) if go is None else go
    torch.autograd.grad(o, qkv, go)
del qkv, o, go
tiles = (torch.rand(rows_mb, 19, 60, device=dev, generator=g) < 0.1).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(REPS): nn_kernels.tile_encoder_forward(te, tiles)
torch.cuda.synchronize()
print("pmc learner workload done")
