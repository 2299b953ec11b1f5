"""Diagnostics (GPU box): k_step with 64 / 32 / 16 games per wave (catan_set_step_wave_games) - parity with the CPU oracle on a
small batch, then deferred and lock-step throughput and the per-kernel HIP-event durations at 65 536 games."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib
from settlers_of_catan_rl_amd.env import VecCatanEnv

for G in (64, 32, 16):
    env = VecCatanEnv(1024, seed=3); env.set_step_wave_games(G)
    ob = oracle_lib.OracleBatch(1024, 3)
    env.random_rollout(0, 1500)
    want = ob.run_random(1500)
    ok = np.array_equal(env.export_state().cpu().numpy(), want) and np.array_equal(env.get_action_masks().cpu().numpy(), ob.masks())
    print(f"G={G}: lock-step parity 1024 x 1500: {ok}", flush=True)
    assert ok
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(8192, 32)
step = 1 << 20
for rep in range(2):
    for G in (64, 32, 16):
        env.set_step_wave_games(G)
        env.random_rollout_deferred(256, 32)
        c0 = int(env.policy_counters().sum()); torch.cuda.synchronize(); t0 = time.perf_counter()
        env.random_rollout_deferred(8192, 32)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c1 = int(env.policy_counters().sum())
        env.random_rollout(step, 64); step += 64
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.random_rollout(step, 2048); step += 2048
        torch.cuda.synchronize(); dl = time.perf_counter() - t0
        kd = env.random_rollout_timed(step, 512, 32); step += 512
        kl = env.random_rollout_timed(step, 512, 0); step += 512
        print(f"G={G:2d}: deferred {(c1 - c0) / dt / 1e6:7.1f} M/s {dt / 8192 * 1e6:6.1f} us/pass active {(c1 - c0) / 8192 / 65536:.3f} | "
              f"lock-step {65536 * 2048 / dl / 1e6:6.1f} M/s {dl / 2048 * 1e6:6.1f} us/step | k_step deferred {kd['k_step'] / 512 * 1e3:5.1f} us, "
              f"lock-step {kl['k_step'] / 512 * 1e3:5.1f} us; sample {kd['k_sample_random'] / 512 * 1e3:5.1f} us", flush=True)
print("invalid", env.invalid_action_count())
