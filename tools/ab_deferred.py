"""Diagnostics: the deferred loop in both forms (CATAN_DEFERRED_FUSED=0/1 at creation), us per pass, executed steps per second, per-kernel event times.
TAG / WINDOW / BUDGET from the environment."""
import time, torch, sys, os
sys.path.insert(0, "/root/repo")
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
W = int(os.environ.get("WINDOW", "32"))
if os.environ.get("BUDGET"):
    env.set_lr_budgets(16, int(os.environ["BUDGET"]))
env.random_rollout_deferred(8192, W)
torch.cuda.synchronize()
c0 = int(env.policy_counters().sum())
t0 = time.perf_counter()
env.random_rollout_deferred(8192, W)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
c1 = int(env.policy_counters().sum())
k = env.random_rollout_timed(0, 512, W)
print(os.environ.get("TAG"), "W", W, "budget", os.environ.get("BUDGET"), "us/pass %.2f" % (dt / 8192 * 1e6), "G steps/s %.3f" % ((c1 - c0) / dt / 1e9), "active %.4f" % ((c1 - c0) / 8192 / 65536), {a: round(b * 1e3 / 512, 2) for a, b in k.items()})
