"""Tunes the library GEMMs of the policy net with PyTorch TunableOp (rocBLAS / hipBLASLt solution search per shape) at the
rollout width and the minibatch width, and writes the result file that settlers_of_catan_rl_amd.policy loads when present."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tun
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_results.csv"
B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, a, _ = net.act(f, lists, lens, masks)

def step(n):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, lp, ent = net.evaluate_actions(f[:n], lists[:n], lens[:n], masks[:n], a[:n])
    loss = v.float().mean() + lp.float().mean() - 0.01 * ent
    opt.zero_grad(); loss.backward(); opt.step()

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def act():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net.act(f, lists, lens, masks)

print("before: train step %.1f ms, act %.1f ms" % (timeit(lambda: step(B)), timeit(act)))
# the widths the pipeline uses: rollout / evaluation passes (65 536 rows), PPO minibatches of config 3 (204 800 rows: the
# 65 536 games are tiled to that many rows), value re-evaluation chunks (262 144 rows)
MB = 204800
rep = -(-MB // B)
fm, lm, nm, mm, am = (t.repeat((rep,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks, a))

def step_mb():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, lp, ent = net.evaluate_actions(fm, lm, nm, mm, am)
    loss = v.float().mean() + lp.float().mean() - 0.01 * ent
    opt.zero_grad(); loss.backward(); opt.step()

VC = 262144
fv, lv, nv = (t.repeat((4,) + (1,) * (t.dim() - 1))[:VC] for t in (f, lists, lens))

def value():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net.get_value(fv, lv, nv)

print("before: minibatch step %.1f ms, value chunk %.1f ms" % (timeit(step_mb, 3), timeit(value, 3)))
tun.enable(True); tun.tuning_enable(True); tun.set_filename(out)
tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
t0 = time.perf_counter()
step(B); act(); step_mb(); value()
torch.cuda.synchronize(); print("tuning took %.0f s" % (time.perf_counter() - t0))
tun.tuning_enable(False)
print("after: train step %.1f ms, act %.1f ms" % (timeit(lambda: step(B)), timeit(act)))
print("after: minibatch step %.1f ms, value chunk %.1f ms" % (timeit(step_mb, 3), timeit(value, 3)))
print("results:", len(tun.get_results()))
