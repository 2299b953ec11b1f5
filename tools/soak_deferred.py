"""Soak test of the deferred schedules (multi-stream): long rollouts at 65 536 games with bit-exact oracle parity of ALL games after every
leg (each game after exactly its own number of decisions; the oracle batch runs on all host threads), plus the conservation invariants.
usage: soak_deferred.py [legs] [passes per leg] [fused 0|1]     (environment switches of include/catan_hip_tuning.h apply)"""
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
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
fused = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
env = VecCatanEnv(n, seed=seed)
env.set_deferred_fused(fused)
ob = oracle_lib.OracleBatch(n, seed)
done = np.zeros(n, dtype=np.int64)
print(f"soak: {n} games, {legs} legs of {iters} passes, fused-sampling loop {fused}, switches {({k: v for k, v in os.environ.items() if k.startswith('CATAN_')})}", flush=True)
for leg in range(legs):
    window = (32, 16, 5, 1)[leg % 4]
    t0 = time.perf_counter()
    env.random_rollout_deferred(iters, window)
    cnt = env.policy_counters().cpu().numpy()
    dt = time.perf_counter() - t0
    blobs = env.export_state().cpu().numpy()
    tot = spec.state_field(blobs, "bank_res").astype(np.int64)
    for p in (1, 2, 3, 4):
        tot = tot + spec.state_field(blobs, f"p{p}_res")
    assert (tot == 19).all() and env.invalid_action_count() == 0
    t1 = time.perf_counter()
    want = ob.run_random_counts(cnt - done, start=done, n_threads=0)
    bad = np.flatnonzero((want != blobs).any(axis=1))
    masks_equal = bool(np.array_equal(env.get_action_masks().cpu().numpy(), ob.masks()))
    for i in bad[:3]:
        print("MISMATCH game", int(i), spec.describe_state_diff(want[i], blobs[i]))
    print(f"leg {leg}: window {window}, {iters} passes in {dt:.2f} s ({(cnt - done).sum() / dt / 1e9:.3f} G env-steps/s), {cnt.sum() / 1e9:.3f} G steps total, "
          f"{ob.games.value} games ended so far; ALL {n} games compared with the oracle ({time.perf_counter() - t1:.0f} s): {len(bad)} differ, masks equal: {masks_equal}", flush=True)
    done = cnt.copy()
    assert len(bad) == 0 and masks_equal
print("soak ok")
