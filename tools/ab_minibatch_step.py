"""Diagnostics (GPU box): A/B of the minibatch step at config 3's real shapes inside ONE process (run-to-run differences between
boxes are larger than the effects looked for): the first `--steps` optimiser steps of an epoch, with the trainer's switches toggled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T, STEPS = 65536, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 10
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
class Stop(Exception): pass
def run(dedupe):
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=64), autocast_dtype=torch.bfloat16, seed=3)
    tr.dedupe_boards = dedupe
    calls = [0]; orig = tr.optimiser.step; t = {}
    def step(*a, **k):
        r = orig(*a, **k); calls[0] += 1
        if calls[0] == 3: torch.cuda.synchronize(); t["a"] = time.perf_counter()
        if calls[0] == 3 + STEPS: torch.cuda.synchronize(); t["b"] = time.perf_counter(); raise Stop()
        return r
    tr.optimiser.step = step
    try: tr.update(st)
    except Stop: pass
    return (t["b"] - t["a"]) / STEPS * 1e3
for rnd in range(3):
    for dedupe in (True, False):
        print(f"round {rnd}: dedupe_boards={dedupe}: {run(dedupe):.2f} ms per minibatch step", flush=True)
