#!/usr/bin/env python
"""bench.py - Catan env-steps/s at 65 536 parallel games per GPU (BASELINE.json configs[1]).

A "step" (--steps) is one pass of the hot path over one batch of games: a uniform-random legal action per game (device
sampler standing in for the policy), EnvWrapper.step semantics (apply + done/reward + auto-reset) and the next
legal-action masks, all inputs resident in HBM.  Two schedules of the same kernels (DESIGN.md "Schedules"):

  --window W (default 32)  deferred loop, catan_random_rollout_deferred: the few games whose step needs the slow path
                           (longest-road search, re-deal after a win) sit out while it runs on side streams; every game
                           still follows its own lock-step trajectory bit for bit (tests/test_gpu_env_parity.py).
                           `value` counts the env steps actually executed (sum of the per-game decision counters).
  --window 0               lock-step loop, catan_random_rollout: every game steps in every pass (the per-step API
                           schedule); also measured briefly in every default run and reported as `lockstep`.

    python bench.py --gpus 1 --steps 4096 --warmup 256
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel of the timed loop, HIP events on the
stream it is launched on) and `cpu_baseline` (the CPU oracle - a port of the reference algorithm - timed on this box's
host cores).  Games shard across ranks by global game id with no data-path collective ("weak": 65 536 games per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per game per launch of each kernel (game-major layout of csrc/catan_state.h; derivation in
# DESIGN.md "Roofline"): the minimum live set an ideal kernel must move, not what the kernel happens to touch.
ALGO_BYTES = {
    "k_sample_random": 44 + 5 + 72 + 4,             # packed masks + own hand + action out + the sort's list entry
    # fused step: action in (72) + packed masks in/out (44 + 44) + reward/done out (17) + the HOT record read
    # (112 words x 4 B = 448) + the part of it that an ideal kernel must write back (hands, estimates, control block,
    # one bitboard word: ~40 words x 4 B = 160)
    "k_step": 72 + 44 + 44 + 17 + 448 + 160,
    "k_classify": 0,                                # (the sort for caller-supplied actions; the rollout loops sort in the sampler)
    "k_lr_finish": 0,                               # slow path (a few % of the games): latency-bound searches / re-deals,
    "k_lr_heavy": 0,                                # no meaningful byte roofline
    "k_step_finish": 0,
    "k_reset_list": 0,
}
FAST_PATH = ("k_sample_random", "k_classify", "k_step")     # the kernels on the timed loop's critical path
HBM_PEAK_GBS = 8000.0                               # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")   # rocprofv3 --pmc passes (tools/profile_round.sh)


def cpu_baseline(sample_envs=16384, sample_steps=2048):
    """The CPU oracle (port of the reference algorithm, bit-identical to the reference by the fixtures) on the host
    cores of this box, bounded sample of the same workload (same seeds/sampler)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build()
    cores = os.cpu_count() or 1
    b = oracle_lib.OracleBatch(sample_envs, seed=0)
    b.run_random(64, want_blobs=False, n_threads=cores)          # warm-up / page-in
    t0 = time.perf_counter()
    b.run_random(sample_steps, want_blobs=False, n_threads=cores)
    dt_all = time.perf_counter() - t0
    one = oracle_lib.OracleBatch(max(64, sample_envs // 32), seed=0)
    t0 = time.perf_counter()
    one.run_random(sample_steps, want_blobs=False, n_threads=1)
    dt_one = time.perf_counter() - t0
    return {
        "value": sample_envs * sample_steps / dt_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{sample_envs} games x {sample_steps} random-policy steps (step+masks, auto-reset), OpenMP over {cores} threads",
        "single_thread_value": one.n * sample_steps / dt_one,
        "reference_python_note": "reference Python EnvWrapper: ~2.1-2.4k env-steps/s/core (BASELINE.md, survey container)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--envs", type=int, default=65536, help="games per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-validate", action="store_true", help="skip the mask-bit legality check in k_step")
    ap.add_argument("--window", type=int, default=32,
                    help="W > 0: deferred loop, tier-2 longest road + re-deals once per W passes; 0: lock-step loop")
    ap.add_argument("--no-lockstep", action="store_true", help="skip the short lock-step measurement of a deferred run")
    args = ap.parse_args()

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the HIP path has no CPU fallback)")
    from settlers_of_catan_rl_amd import dist as cdist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    # backend "nccl" == RCCL over xGMI; CATAN_DIST_BACKEND=gloo lets two ranks share one GPU (smoke test of this code path)
    rank, local_rank, world = cdist.init_from_env(backend=os.environ.get("CATAN_DIST_BACKEND") or None)

    from settlers_of_catan_rl_amd.env import VecCatanEnv

    env_id0, n = cdist.shard(rank, args.envs)                # global game ids: results do not depend on `world`
    env = VecCatanEnv(n, seed=args.seed, env_id0=env_id0, validate_actions=not args.no_validate, auto_reset=True)

    if args.window > 0:
        env.random_rollout_deferred(args.warmup, args.window)
        c0 = int(env.policy_counters().sum())
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout_deferred(args.steps, args.window)
        cdist.barrier()
        dt = cdist.max_over_ranks(time.perf_counter() - t0)     # max over ranks
        my_steps = int(env.policy_counters().sum()) - c0        # decisions actually executed by this rank's games
        env_steps = cdist.sum_over_ranks(my_steps)
    else:
        env.random_rollout(0, args.warmup)
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout(args.warmup, args.steps)
        cdist.barrier()
        dt = cdist.max_over_ranks(time.perf_counter() - t0)     # max over ranks
        my_steps = n * args.steps
        env_steps = world * n * args.steps
    bad = env.invalid_action_count()

    lockstep = None
    if args.window > 0 and not args.no_lockstep:
        # the same games, continued in lock-step for a short stretch (every game steps in every pass)
        ls_steps = min(args.steps, 512)
        env.random_rollout(1 << 20, 32)
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout((1 << 20) + 32, ls_steps)
        cdist.barrier()
        ls_dt = cdist.max_over_ranks(time.perf_counter() - t0)
        lockstep = {"value": world * n * ls_steps / ls_dt, "unit": "env-steps/s", "steps": ls_steps, "ms_per_step": ls_dt / ls_steps * 1e3}

    out = None
    if rank == 0:
        # per-kernel durations: HIP events on the stream each kernel is launched on, a separate short pass right after
        prof_steps = min(args.steps, 512)
        kms = env.random_rollout_timed(args.warmup + args.steps, prof_steps, args.window)
        slow_launches = prof_steps if args.window <= 0 else -(-prof_steps // args.window)
        launches = {k: (slow_launches if k in ("k_lr_heavy", "k_step_finish", "k_reset_list") else prof_steps) for k in kms}
        per_launch_us = {k: v * 1e3 / launches[k] for k, v in kms.items() if k in ALGO_BYTES}
        dom = max(FAST_PATH, key=per_launch_us.get)
        active = my_steps / (n * args.steps)                  # games that take a step in a pass (the others are busy)
        achieved = ALGO_BYTES[dom] * n * active / (per_launch_us[dom] * 1e-6) / 1e9
        fast_us = sum(per_launch_us[k] for k in FAST_PATH)
        traffic = None
        if os.path.exists(PMC_SUMMARY):
            with open(PMC_SUMMARY) as f:
                traffic = json.load(f).get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": ("profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated; "
                               "65 536 games, same kernels)") if traffic is not None else None,
            "algorithmic_bytes_per_game": ALGO_BYTES[dom],
            "algorithmic_bytes_per_launch": ALGO_BYTES[dom] * n * active,
            "avg_launch_us": per_launch_us[dom],
            "all_kernels_avg_launch_us": per_launch_us,
            "fast_path_algorithmic_bytes_per_game": sum(ALGO_BYTES[k] for k in FAST_PATH),
            "fast_path_achieved_gbs": sum(ALGO_BYTES[k] for k in FAST_PATH) * n * active / (fast_us * 1e-6) / 1e9,
            "note": "integer/byte rules engine at one wave per SIMD: latency- and divergence-bound, far below the HBM "
                    "roofline by nature (SURVEY.md 8(d)); frac is reported for the dominant kernel of the timed loop; "
                    "the slow-path kernels (k_lr_*, k_step_finish, k_reset_list) run on side streams in the deferred loop",
        }
        value = env_steps / dt
        out = {
            "metric": "Catan env-steps/sec at 65k parallel games", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 (integer bitboards)",
            "data": "synthetic (random-seed boards, uniform-random legal policy on device)",
            "config": {"workload": "configs[1]: 65 536 parallel envs per GPU, random policy, step+mask only, "
                                   "bit-exact vs CPU oracle", "games_per_gpu": n, "validate_actions": not args.no_validate,
                       "auto_reset": True, "parallelism": f"games sharded over {world} GPU(s), no collective",
                       "schedule": (f"deferred, window {args.window}: slow-path games sit out; value = executed env steps / time"
                                    if args.window > 0 else "lock-step: every game steps in every pass")},
            "env_steps_executed": env_steps, "active_fraction": env_steps / (world * n * args.steps),
            "invalid_actions": bad,
            "roofline": roofline,
        }
        if lockstep is not None:
            out["lockstep"] = lockstep
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
    if world > 1:
        cdist.finalize()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
