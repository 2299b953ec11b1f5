"""Diagnostics (GPU box): microseconds per pass of the deferred random-policy loop at 65 536 games under the process's environment
(CATAN_STEP_WAVES_PER_BLOCK, CATAN_DEFERRED_FUSED, CATAN_STEP_WAVE_GAMES ...; WINDOW = the deferred window), and the per-kernel HIP-event durations.  One line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
if os.environ.get("MAIN_PRIO"):                          # the loop's own stream with a stream priority (-1: high)
    torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["MAIN_PRIO"])))
env = VecCatanEnv(65536, seed=0)
W = int(os.environ.get("WINDOW", "32"))
if os.environ.get("LOCK_BUDGET"):                        # tier-1 search budget inside a lock-step step (default 16)
    env.set_lr_budgets(int(os.environ["LOCK_BUDGET"]), 12)
if os.environ.get("LR_BUDGET"):                          # tier-1 search budget of the deferred schedules (default 12)
    env.set_lr_budgets(16, int(os.environ["LR_BUDGET"]))
env.random_rollout_deferred(8192, W)
out = []
for rep in range(3):
    env.random_rollout_deferred(256, W)
    c0 = int(env.policy_counters().sum()); torch.cuda.synchronize(); t0 = time.perf_counter()
    env.random_rollout_deferred(8192, W)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c1 = int(env.policy_counters().sum())
    out.append((round(dt / 8192 * 1e6, 2), round((c1 - c0) / dt / 1e6, 1), round((c1 - c0) / 8192 / 65536, 4)))
kd = env.random_rollout_timed(1 << 20, 512, W)
env.random_rollout(1 << 21, 64)
torch.cuda.synchronize(); t0 = time.perf_counter()
env.random_rollout(1 << 22, 1024)
torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / 1024 * 1e6
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("CATAN_") or k in ("MAIN_PRIO", "LR_BUDGET", "LOCK_BUDGET", "WINDOW")}, "us_per_pass / M steps per s / active": out,
                  "k_step_us": round(kd["k_step"] / 512 * 1e3, 2), "k_sample_random_us": round(kd["k_sample_random"] / 512 * 1e3, 2),
                  "lockstep_us_per_step": round(dl, 1), "invalid": env.invalid_action_count()}), flush=True)
