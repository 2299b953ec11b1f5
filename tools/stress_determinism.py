"""GPU box: run-to-run bit-stability of the kernels whose instruction streams were rebuilt in round 4 (fused tile encoder - inference and
training forward -, attention forward / backward) at full width with two or more workgroups per CU: every repetition must equal the
first, and a batch of repeated boards must equal the small batch repeated (the packed-f32 instability showed up exactly
there).  Prints one line per kernel; exit code 1 on a difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import policy as P, nn_kernels
from settlers_of_catan_rl_amd.env import VecCatanEnv

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
net = P.CatanPolicy().cuda()
with torch.no_grad():
    for p in net.parameters():
        p.add_(0.05 * torch.randn_like(p))
te = net.observation_module.tile_encoder
env = VecCatanEnv(4099, seed=3); env.random_rollout(0, 600)
f, _, _ = env.get_obs()
small = f[:, 18:18 + 1140].reshape(-1, 19, 60).to(torch.bfloat16).contiguous()
big = small.repeat(50, 1, 1)[:204800].contiguous()
bad = 0
with torch.no_grad():
    ref_small = nn_kernels.tile_encoder_forward(te, small)
    ref = ref_small.repeat(50, 1)[:204800]
    diff = sum(int((nn_kernels.tile_encoder_forward(te, big) != ref).any(1).sum()) for _ in range(REPS))
    print(f"k_tile_encoder_fwd<false>: {REPS} runs of 204 800 boards, rows differing from the 4 099-board run repeated: {diff}", flush=True); bad += diff
with torch.autocast("cuda", dtype=torch.bfloat16):
    outs = [nn_kernels.tile_encoder_train(te, big[:65536]) for _ in range(2)]
    first = outs[0].detach().clone()
    diff = 0
    for _ in range(max(REPS // 3, 3)):
        o = nn_kernels.tile_encoder_train(te, big[:65536])
        diff += int((o.detach() != first).any(1).sum())
    print(f"k_tile_encoder_fwd<true>: rows differing between runs of 65 536 boards: {diff}", flush=True); bad += diff
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(204800, 19, 3, 4, 16, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
o0 = nn_kernels.small_attention(qkv); go = torch.randn_like(o0)
g0, = torch.autograd.grad(o0, qkv, go, retain_graph=True)
df = db = 0
for _ in range(REPS):
    o = nn_kernels.small_attention(qkv)
    df += int((o != o0).any(-1).sum())
    gq, = torch.autograd.grad(o, qkv, go)
    db += int((gq != g0).flatten(1).any(1).sum())
print(f"k_attn_mfma_fwd / _bwd: {REPS} runs of 204 800 sequences, differing rows {df} / sequences {db}", flush=True); bad += df + db
# ---- round 5: the other kernels that pack bf16 pairs with v_cvt_pk_bf16_f32 - the row products, the LayerNorms of every width
# (forward and the dX of their backward), the row gathers / per-board sums, the policy pass (tile encoder + heads + value, deterministic
# arg-max actions) - each against its own first run.  (Weight gradients are fp32 atomic sums: their order is free by design, not checked.)
def stable(name, fn, reps=REPS):
    global bad
    first = [t.detach().clone() for t in fn()]
    diff = 0
    for _ in range(reps):
        diff += sum(int((a != b).reshape(a.shape[0], -1).any(1).sum()) for a, b in zip(fn(), first))
    print(f"{name}: {reps} runs, differing rows {diff}", flush=True); bad += diff

rows = 19 * 204800
x64 = torch.randn(rows, 64, device="cuda", generator=g).to(torch.bfloat16)
w128 = torch.randn(128, 64, device="cuda", generator=g).to(torch.bfloat16) * 0.1
b128 = torch.randn(128, device="cuda", generator=g).to(torch.bfloat16)
stable("k_linear_rows 64 -> 128 (3.9 M rows)", lambda: [nn_kernels._linear_rows(x64, w128, b128)], max(REPS // 3, 3))
for width, nrows in ((64, rows), (128, 614400), (256, 614400), (512, 204800)):
    ln = torch.nn.LayerNorm(width).cuda()
    xs = torch.randn(nrows, width, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(nrows, width, device="cuda", generator=g).to(torch.bfloat16)
    def fwd_bwd(ln=ln, xs=xs, gy=gy):
        y = nn_kernels.small_layer_norm(xs, ln, relu=True)
        gx, = torch.autograd.grad(y, xs, gy)
        return [y, gx]
    if nn_kernels.ln_supported(xs, ln):
        stable(f"LayerNorm {width} forward + dX ({nrows} rows)", fwd_bwd, max(REPS // 3, 3))
src = torch.randn(179200, 480, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
inv = torch.randint(0, 179200, (204800,), device="cuda", generator=g)
order = torch.argsort(inv); start = torch.searchsorted(inv[order], torch.arange(179201, device="cuda"))
gy = torch.randn(204800, 480, device="cuda", generator=g).to(torch.bfloat16)
def expand():
    y = nn_kernels.expand_rows(src, inv, order, start)
    gs, = torch.autograd.grad(y, src, gy)
    return [y, gs]
stable("k_expand_rows16 + k_segment_sum16 (179 200 -> 204 800 rows of 480)", expand, max(REPS // 3, 3))
env2 = VecCatanEnv(65536, seed=5); env2.random_rollout(0, 700)
of, ol, on = env2.get_obs_rows(torch.bfloat16)
mk = env2.get_action_masks()
shadow = net.inference_copy(torch.bfloat16)
def act():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        v, a, lp = shadow.act(of, ol, on, mk, deterministic=True)[:3]
    return [v.float(), a, lp.float()]
stable("policy pass at 65 536 rows (k_tile_encoder_fwd, row kernels, k_head_fwd x 18, value head), arg-max actions", act, max(REPS // 3, 3))
sys.exit(1 if bad else 0)
