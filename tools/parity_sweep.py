"""GPU box: a broader bit-exactness sweep than the test-suite runs - seeds x configurations (dense rewards, trade limits,
validation on/off), lock-step and deferred schedules, against the CPU oracle.  Prints one line per case; exit code 1 on a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle_lib
from settlers_of_catan_rl_amd.env import VecCatanEnv

bad = 0
big = [(606, 65536, 2000, {}), (707, 32768, 4000, dict(dense_reward=True))] if "--big" in sys.argv else []
for seed, n, steps, kw in big + [(101, 2048, 1500, {}), (202, 1000, 2500, dict(dense_reward=True)), (303, 3000, 1200, dict(max_proposed_trades_per_turn=1)),
                           (404, 513, 3000, dict(validate_actions=False)), (505, 4096, 1000, dict(max_proposed_trades_per_turn=7, dense_reward=True))]:
    okw = {k: v for k, v in kw.items() if k in ("dense_reward", "max_proposed_trades_per_turn")}
    # lock-step
    env = VecCatanEnv(n, seed=seed, **kw)
    ob = oracle_lib.OracleBatch(n, seed)
    ob.set_config(max_trades_per_turn=okw.get("max_proposed_trades_per_turn", 4), dense_reward=okw.get("dense_reward", False))
    env.random_rollout(0, steps)
    o = ob.run_random(steps, n_threads=0 if n < 8192 else (os.cpu_count() or 1))
    same = np.array_equal(env.export_state().cpu().numpy(), o) and np.array_equal(env.get_action_masks().cpu().numpy(), ob.masks())
    print(f"lock-step seed {seed} n {n} steps {steps} {kw}: {'OK' if same else 'MISMATCH'} (games finished {ob.games.value})", flush=True)
    bad += not same
    # deferred: every game is compared after exactly the decisions it took
    for window in (8, 32):
        env = VecCatanEnv(n, seed=seed, **kw)
        ob = oracle_lib.OracleBatch(n, seed)
        ob.set_config(max_trades_per_turn=okw.get("max_proposed_trades_per_turn", 4), dense_reward=okw.get("dense_reward", False))
        env.random_rollout_deferred(steps, window)
        cnt = env.policy_counters().cpu().numpy()
        o = ob.run_random_counts(cnt, n_threads=0 if n < 8192 else (os.cpu_count() or 1))
        same = np.array_equal(env.export_state().cpu().numpy(), o) and np.array_equal(env.get_action_masks().cpu().numpy(), ob.masks())
        print(f"deferred W={window} seed {seed}: {'OK' if same else 'MISMATCH'} ({int(cnt.sum())} decisions)", flush=True)
        bad += not same
    assert env.invalid_action_count() == 0
    # catan_step_deferred with caller-supplied actions (round 4): the oracle batch is the policy stub and the shadow env; every
    # delivered reward / done of every call and the flushed states are compared
    calls = min(steps, 300 if n >= 32768 else 600)        # (every call is a host round trip through the oracle batch: bounded)
    for window in (4, 32):
        env = VecCatanEnv(n, seed=seed, **kw)
        ob = oracle_lib.OracleBatch(n, seed)
        ob.set_config(max_trades_per_turn=okw.get("max_proposed_trades_per_turn", 4), dense_reward=okw.get("dense_reward", False))
        counts = np.zeros(n, dtype=np.uint32); waiting = np.zeros(n, dtype=bool)
        acts = np.zeros((n, 18), dtype=np.int32)
        er = np.zeros((n, 4), dtype=np.float32); er64 = np.zeros((n, 4), dtype=np.float64); ed = np.zeros(n, dtype=np.uint8)
        same, applied = True, 0
        for t in range(calls):
            applied += ob.play(counts, (~waiting).astype(np.uint8), acts, er, er64, ed)
            rew, done, status = env.step_deferred(torch.from_numpy(acts).cuda(), window)
            s_ = status.cpu().numpy(); w = s_ == 1
            r_, d_ = rew.cpu().numpy(), done.cpu().numpy()
            same &= not r_[w].any() and not d_[w].any() and np.array_equal(r_[~w], er[~w]) and np.array_equal(d_[~w], ed[~w])
            waiting = w
        rew, done, status = env.step_flush()
        r_, d_ = rew.cpu().numpy(), done.cpu().numpy()
        same &= np.array_equal(r_[waiting], er[waiting]) and np.array_equal(d_[waiting], ed[waiting])
        same &= np.array_equal(env.export_state().cpu().numpy(), ob.export()) and np.array_equal(env.get_action_masks().cpu().numpy(), ob.masks())
        same &= env.invalid_action_count() == 0
        print(f"catan_step_deferred W={window} seed {seed} n {n} calls {calls} {kw}: {'OK' if same else 'MISMATCH'} ({applied} applied actions from outside the library)", flush=True)
        bad += not same
sys.exit(1 if bad else 0)
