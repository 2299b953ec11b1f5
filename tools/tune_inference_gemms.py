"""TunableOp search for the library GEMMs of the INFERENCE passes at the row counts they really run at: the rollout's policy pass at
every captured bucket width (N, 3N/4, N/2, ... 1 024 rows) and the value re-evaluation's chunks (1 048 576 rows and the last, shorter
chunk of a T = 200 rollout; the opponents' module sees three times as many).  Writes the TunableOp csv to argv[1]; the lines are
merged into settlers_of_catan_rl_amd/tunableop_gfx950.csv (tools/merge_tunableop.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tun
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
from settlers_of_catan_rl_amd import nn_kernels

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_inference.csv"
N, T = 65536, int(os.environ.get("T", "200"))
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
buckets = col._bucket_list() if not os.environ.get('VALUES_ONLY') else ()
st = col.gather_rollouts()
tr = PPOTrainer(net, PPOConfig(), autocast_dtype=torch.bfloat16, seed=3)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


inf = net.inference_copy(torch.bfloat16)
f, lists, lens = env.get_obs_rows(torch.bfloat16); masks = env.get_action_masks()
g = torch.Generator(device="cuda").manual_seed(0)


def act(b):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        inf.act(f[:b], lists[:b], lens[:b], masks[:b], generator=g)


before = {"values": timed(lambda: tr.compute_values(st))}
for b in buckets:
    before[b] = timed(lambda: act(b))
tun.enable(True); tun.tuning_enable(True); tun.set_filename(out)
tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "300"))); tun.set_max_tuning_iterations(30)
t0 = time.perf_counter()
tr.compute_values(st)
for b in buckets:
    act(b)
torch.cuda.synchronize(); print("tuning took %.0f s" % (time.perf_counter() - t0), flush=True)
tun.tuning_enable(False)
print("compute_values: %.1f -> %.1f ms" % (before["values"], timed(lambda: tr.compute_values(st))))
for b in buckets:
    print("act at %6d rows (eager): %.2f -> %.2f ms" % (b, before[b], timed(lambda: act(b))))
