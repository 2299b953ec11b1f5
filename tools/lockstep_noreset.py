"""Diagnostics (GPU box): lock-step step time with and without the re-deal of finished games (auto_reset off: a state
imported from a steady-state run, finished games stay finished) - how much of a lock-step step is the re-deal's tail."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
n = 65536
env = VecCatanEnv(n, seed=0)
env.random_rollout(0, 3000)
blob = env.export_state()
for auto in (True, False):
    e2 = VecCatanEnv(n, seed=0, auto_reset=auto)
    e2.import_state(blob)
    e2.random_rollout(3000, 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e2.random_rollout(3008, 128)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 128
    k = e2.random_rollout_timed(3136, 64, 0)
    print(f"auto_reset={auto}: {dt * 1e6:.1f} us per step; " + ", ".join(f"{a} {b / 64 * 1e3:.0f}" for a, b in k.items()))
e3 = VecCatanEnv(n, seed=0); e3.import_state(blob); e3.random_rollout(3000, 600)
print("finished games re-dealt on the critical path for lack of a speculative successor:", e3.missed_speculation_count())
