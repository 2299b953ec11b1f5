"""Diagnostics (GPU box): two rollout + update rounds as bench.py's ppo_update runs them, with the trainer's timings per round and
the share of distinct boards per minibatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T = 65536, 200
env = VecCatanEnv(N, seed=0)
pre = int(os.environ.get("PREROLL", "500"))
env.random_rollout(0, pre)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=0, autocast_dtype=torch.bfloat16)
tr = PPOTrainer(net, PPOConfig(ppo_epoch=int(os.environ.get("EPOCHS", "2"))), autocast_dtype=torch.bfloat16, seed=0)
orig = tr.minibatch_boards
def mb(*a, **k):
    r = orig(*a, **k)
    print("   distinct boards per minibatch: %.3f of the rows" % (sum(x[0].numel() for x in r) / (len(r) * r[0][1].numel())), flush=True)
    return r
tr.minibatch_boards = mb
for u in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = col.gather_rollouts(); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.update(st); torch.cuda.synchronize(); t2 = time.perf_counter()
    col.after_rollouts()
    tm = tr.timings
    ms = torch.cuda.memory_stats()
    print("   allocator: reserved %.1f GB, allocated peak %.1f GB, hipMalloc calls %d, retries %d, segments freed %d" % (
        ms["reserved_bytes.all.current"] / 2**30, ms["allocated_bytes.all.peak"] / 2**30, ms["segment.all.allocated"], ms["num_alloc_retries"], ms["segment.all.freed"]), flush=True)
    print(f"round {u}: rollout {t1 - t0:.2f} s ({col.iters} passes), update {t2 - t1:.2f} s: values {tm['values_s']:.2f}, minibatches {tm['minibatches_s']:.2f} = {tm['minibatches_s'] / (tr.cfg.ppo_epoch * 64) * 1e3:.2f} ms per step", flush=True)
