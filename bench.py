#!/usr/bin/env python
"""bench.py - Catan env-steps/s at 65 536 parallel games per GPU (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch: for every game of this rank, one uniform-random legal
action (device sampler standing in for the policy), EnvWrapper.step semantics (apply + done/reward + auto-reset)
and the next legal-action masks - i.e. k_sample_random -> k_step (fused) -> k_lr_heavy -> k_step_finish, all inputs
resident in HBM.

    python bench.py --gpus 1 --steps 4096 --warmup 256
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel, HIP events on the launch stream)
and `cpu_baseline` (the CPU oracle - a port of the reference algorithm - timed on this box's host cores).
Games shard across ranks by global game id with no data-path collective ("weak" scaling: 65 536 games per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per game per launch of each kernel (packed layout of csrc/catan_state.h; derivation in
# DESIGN.md "Roofline"): the minimum live set an ideal kernel must move, not what the kernel happens to touch.
ALGO_BYTES = {
    "k_sample_random": 44 + 5 + 72,                 # packed masks + own hand + action out
    # fused step: action in (72) + packed masks in/out (44 + 44) + reward/done out (17) + the HOT state tile read
    # (112 rows x 4 B = 448) + the part of it that an ideal kernel must write back (hands, estimates, control block,
    # one bitboard word: ~40 rows x 4 B = 160)
    "k_step": 72 + 44 + 44 + 17 + 448 + 160,
    "k_classify": 4 + 4,                            # action type in, permutation out
    "k_lr_finish": 0,                                    # longest-road tiers: LDS/ALU only (3 bitboard words per request)
    "k_lr_heavy": 0,
    "k_step_finish": 0,                             # completes the ~3 % of games that placed a road / settlement
    "k_reset_list": 0,                              # ~0.1 % of games per step end and are re-dealt
}
HBM_PEAK_GBS = 8000.0                               # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


def cpu_baseline(sample_envs=2048, sample_steps=1024):
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
    one = oracle_lib.OracleBatch(max(64, sample_envs // 16), seed=0)
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
    ap.add_argument("--window", type=int, default=0,
                    help="0: lock-step loop; W > 0: deferred loop, slow path (longest road, re-deal) once per W iterations")
    args = ap.parse_args()

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the HIP path has no CPU fallback)")
    from settlers_of_catan_rl_amd import dist as cdist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, local_rank, world = cdist.init_from_env()          # backend "nccl" == RCCL over xGMI

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
        env_steps = cdist.sum_over_ranks(int(env.policy_counters().sum()) - c0)   # decisions actually executed
    else:
        env.random_rollout(0, args.warmup)
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout(args.warmup, args.steps)
        cdist.barrier()
        dt = cdist.max_over_ranks(time.perf_counter() - t0)     # max over ranks
        env_steps = world * n * args.steps
    bad = env.invalid_action_count()

    out = None
    if rank == 0:
        # per-kernel durations: HIP events on the launch stream, a separate short pass right after the timed region
        prof_steps = min(args.steps, 512)
        kms = env.random_rollout_timed(args.warmup + args.steps, prof_steps, args.window)
        slow_launches = prof_steps if args.window <= 0 else -(-prof_steps // args.window)
        launches = {k: (slow_launches if k in ("k_lr_heavy", "k_step_finish", "k_reset_list") else prof_steps) for k in kms}
        per_launch_us = {k: v * 1e3 / launches[k] for k, v in kms.items() if k in ALGO_BYTES}
        dom = max(per_launch_us, key=per_launch_us.get)
        achieved = ALGO_BYTES[dom] * n / (per_launch_us[dom] * 1e-6) / 1e9
        total_us = sum(per_launch_us.values())
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_game": ALGO_BYTES[dom],
            "avg_launch_us": per_launch_us[dom],
            "all_kernels_avg_launch_us": per_launch_us,
            "whole_step_algorithmic_bytes_per_game": sum(ALGO_BYTES.values()),
            "whole_step_achieved_gbs": sum(ALGO_BYTES.values()) * n / (total_us * 1e-6) / 1e9,
            "note": "integer/byte rules engine: latency- and divergence-bound, far below the HBM roofline by nature "
                    "(SURVEY.md 8(d)); frac is reported for the dominant kernel as the contract asks",
        }
        value = env_steps / dt
        out = {
            "metric": "Catan env-steps/sec at 65k parallel games", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 (integer bitboards)",
            "data": "synthetic (random-seed boards, uniform-random legal policy on device)",
            "config": {"workload": "configs[1]: 65 536 parallel envs per GPU, random policy, step+mask only, "
                                   "bit-exact vs CPU oracle", "games_per_gpu": n, "validate_actions": not args.no_validate,
                       "auto_reset": True, "parallelism": f"games sharded over {world} GPU(s), no collective"},
            "invalid_actions": bad,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
    if world > 1:
        cdist.finalize()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
