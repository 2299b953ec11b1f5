"""The fused-sampling deferred loop's window-close race, made deterministic.

A window of an ODD number of passes closes in the middle of a tier-1 group; until round 6 nothing ordered that pass's k_step before the
window's re-deal kernel on the side stream, so a game that ended in that pass could miss the re-deal list's length being read and was
never re-dealt.  Environment (read by libcatan_hip.so when the first deferred pass runs):
  CATAN_DEBUG_FUSED_CLOSE_UNORDERED=1   the round-4/5 ordering (no dependency on the closing pass's k_step)
  CATAN_DEBUG_STEP_DELAY_US=k           that k_step starts k microseconds late (a spinning one-wave kernel in front of it)
Prints, per window length, the number of games whose state differs from the oracle's after exactly their own number of decisions.
tools/profile_round6.sh runs: unordered + delay (games are lost), ordered + delay (none), ordered (none)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib
from settlers_of_catan_rl_amd.env import VecCatanEnv

label = f"unordered={os.environ.get('CATAN_DEBUG_FUSED_CLOSE_UNORDERED', '0')} delay_us={os.environ.get('CATAN_DEBUG_STEP_DELAY_US', '0')}"
worst = 0
for n, iters, window, seed in ((300, 1500, 1, 5), (2048, 1777, 5, 4), (1024, 1500, 8, 0)):
    env = VecCatanEnv(n, seed=seed)
    env.set_deferred_fused(True)
    ob = oracle_lib.OracleBatch(n, seed)
    env.random_rollout_deferred(iters, window)
    cnt = env.policy_counters().cpu().numpy()
    want = ob.run_random_counts(cnt)
    got = env.export_state().cpu().numpy()
    bad = int((want != got).any(axis=1).sum())
    worst = max(worst, bad)
    print(f"{label}: window {window}, {n} games x {iters} passes: {bad} games differ from the oracle ({ob.games.value} games ended)", flush=True)
print(f"{label}: {'LOST GAMES' if worst else 'all games on their lock-step trajectories'}")
