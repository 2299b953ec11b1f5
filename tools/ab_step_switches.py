"""Diagnostics (GPU box): A/B of the minibatch step at config 3's real shapes inside ONE process, switch by switch (run-to-run
differences between boxes are larger than the effects looked for).  SWITCHES=name,name (default: all); each is toggled off / on in
turn, `--steps` optimiser steps each, three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd import nn_kernels, policy as pol_mod
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector
from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
N, T, STEPS = 65536, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 16
env = VecCatanEnv(N, seed=0); env.random_rollout(0, 500)
net = CatanPolicy().cuda()
col = RolloutCollector(env, net, T, seed=1, autocast_dtype=torch.bfloat16)
st = col.gather_rollouts()
del col


def set_arena(on): nn_kernels.grad_arena.enabled = on
def set_trunk(on): pol_mod.TRUNK_WINDOWS = on


def set_tuned(on):
    import torch.cuda.tunable as tun
    tun.enable(on)


def set_grouped(on): pol_mod.GROUPED_WGRAD = on


def set_deferred(on): nn_kernels.wgrad_queue.enabled = on


def set_recurrent(on): pol_mod.RECURRENT_BATCHED = on
def set_gather_ranges(on): nn_kernels.GATHER_RANGES = on
def set_fanout(on): nn_kernels.FANOUT_GATHER = on
def set_concat(on): nn_kernels.CONCAT_ROWS = on
def set_wgrad_big(on): nn_kernels.WGRAD_BIG = on
def set_recompute_h(on): os.environ["CATAN_TE_RECOMPUTE_H"] = "1" if on else "0"      # (read by every tile-encoder forward)


def set_cat_bits(on): nn_kernels.CATEGORICAL_BITS = on
SW = {"cat_bits": set_cat_bits, "wgrad_big": set_wgrad_big, "recompute_h": set_recompute_h, "grad_arena": set_arena, "tuned_gemms": set_tuned, "grouped_wgrad": set_grouped, "deferred_wgrad": set_deferred, "recurrent_batched": set_recurrent, "gather_ranges": set_gather_ranges, "fanout_gather": set_fanout, "concat_rows": set_concat}
if hasattr(pol_mod, "TRUNK_WINDOWS"):
    SW["trunk_windows"] = set_trunk


class Stop(Exception):
    pass


def run():
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=1, num_mini_batch=64), autocast_dtype=torch.bfloat16, seed=3)
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


DEFAULT_OFF = {"recompute_h"}
names = [n for n in os.environ.get("SWITCHES", ",".join(SW)).split(",") if n in SW]
run()                                                   # warm-up (GEMM kernels, arena size)
for rnd in range(3):
    for n in names:
        for on in (False, True):
            SW[n](on)
            print(f"round {rnd}: {n}={'on ' if on else 'off'}: {run():.2f} ms per minibatch step", flush=True)
        SW[n](n not in DEFAULT_OFF)                      # back to the library's default before the next pair
