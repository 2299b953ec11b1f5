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
sys.exit(1 if bad else 0)
