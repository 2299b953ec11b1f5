"""Diagnostics (GPU box): is a minibatch step of config 3 host-bound or device-bound?  Times STEPS consecutive optimiser steps of a
real update twice: until the host has QUEUED them (no synchronisation) and until the device has finished them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T, STEPS = 65536, int(os.environ.get("T", "200")), int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
class Stop(Exception): pass
for rnd in range(4):
    import torch.cuda.tunable as tun
    tun.enable(rnd % 2 == 0)
    print("TunableOp enabled:", tun.is_enabled())
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=64), autocast_dtype=torch.bfloat16, seed=3)
    calls = [0]; orig = tr.optimiser.step; t = {}
    def step(*a, **k):
        r = orig(*a, **k); calls[0] += 1
        if calls[0] == 5: torch.cuda.synchronize(); t["a"] = time.perf_counter()
        if calls[0] == 5 + STEPS:
            t["q"] = time.perf_counter(); torch.cuda.synchronize(); t["b"] = time.perf_counter(); raise Stop()
        return r
    tr.optimiser.step = step
    try: tr.update(st)
    except Stop: pass
    print(f"round {rnd}: {STEPS} steps queued by the host in {(t['q'] - t['a']) / STEPS * 1e3:.2f} ms per step, finished by the device in {(t['b'] - t['a']) / STEPS * 1e3:.2f} ms per step", flush=True)
