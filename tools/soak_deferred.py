"""Soak test of the deferred schedule (multi-stream): long rollouts at 65 536 games with bit-exact oracle parity on a sample of
the games after every leg (each game after exactly its own number of decisions), plus the conservation invariants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle_lib
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import spec

n, seed = 65536, 12
legs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
env = VecCatanEnv(n, seed=seed)
sample = np.arange(0, n, 1021)
orc = {int(i): oracle_lib.OracleBatch(1, seed, env_id0=int(i)) for i in sample}
done = np.zeros(n, dtype=np.int64)
for leg in range(legs):
    t0 = time.perf_counter()
    env.random_rollout_deferred(iters, 32 if leg % 2 == 0 else 16)
    cnt = env.policy_counters().cpu().numpy()
    dt = time.perf_counter() - t0
    blobs = env.export_state().cpu().numpy()
    tot = spec.state_field(blobs, "bank_res").astype(np.int64)
    for p in (1, 2, 3, 4):
        tot = tot + spec.state_field(blobs, f"p{p}_res")
    assert (tot == 19).all() and env.invalid_action_count() == 0
    bad = 0
    for i in sample:
        want = orc[int(i)].run_random_counts(np.array([cnt[i] - done[i]]), start=np.array([done[i]]))
        if not np.array_equal(want[0], blobs[i]):
            bad += 1
            print("MISMATCH game", i, spec.describe_state_diff(want[0], blobs[i]))
    done = cnt.copy()
    print(f"leg {leg}: {iters} passes in {dt:.2f} s, {(cnt.sum()) / 1e9:.3f} G steps total, sampled games checked {len(sample)}, mismatches {bad}", flush=True)
    assert bad == 0
print("soak ok")
