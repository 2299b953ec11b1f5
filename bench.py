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
    # fused step: action in (72) + packed masks out (44) + reward/done out (17) + the HOT record read (112 words x 4 B = 448) + the
    # part of it that an ideal kernel must write back (40 words x 4 B = 160: counters, control block, two hands, nine estimate
    # words, one bitboard).  static_assert-ed against the layout in csrc/catan_state.h (STEP_ALGO_BYTES) and checked against the
    # library at run time (catan_step_algorithmic_bytes).  Rounds 1-3 also read the previous masks (44): validate mode now
    # restates Game.validate_action from the state.
    "k_step": 72 + 44 + 17 + 448 + 160,
    # the fused-sampling step (the default deferred loop since round 6: no sampler kernel): action + decision counter in from the game's side row
    # (76), hot record in (448), masks + next action + counter out into the side row (120), reward / done out (17), the ideal write-back (160),
    # the game's id out of this pass's list and into the next pass's (8).  STEP_FUSED_ALGO_BYTES in csrc/catan_state.h, checked at run time.
    "k_step_fused": 76 + 448 + 120 + 17 + 160 + 8,
    "k_lr_finish": 0,                               # slow path: priced per REQUEST below (LR_REQUEST_BYTES), not per game
    "k_lr_heavy": 0,
    "k_reset_list": 0,
}
# a longest-road request (~3 % of the games of a pass): hot record in and out (448 + 448), the longest-path cache (28 + 28),
# new masks (44), reward / done (17).  The search itself runs in LDS / registers: the kernels are LATENCY-bound (a DFS step
# is ~1 us of dependent instructions at this occupancy), the byte figure only shows how far from any HBM limit they are.
LR_REQUEST_BYTES = 448 + 448 + 28 + 28 + 44 + 17
FAST_PATH = ("k_sample_random", "k_step")     # the kernels on the timed loop's critical path
HBM_PEAK_GBS = 8000.0                               # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")   # rocprofv3 --pmc passes (tools/profile_round6.sh)


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


def live_pmc_traffic(kernel, timeout_s=180):
    """HBM bytes per launch of `kernel`, MEASURED NOW: the two counter passes MI355X_MICROARCH.md prescribes (rocprofv3 --pmc
    FETCH_SIZE, then --pmc WRITE_SIZE; never combined with a trace domain) over tools/pmc_workload.py - the same 65 536 games
    after the same kind of pre-roll, plus k_calib_copy launches of exactly known traffic that calibrate the counter units -
    summarised by tools/pmc_summarise.py (mean over the last 96 launches).  Returns (bytes, note) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    out = tempfile.mkdtemp(prefix="catan_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(out, sub), "-o", "pmc", "--",
                                sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py")],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} exited with {r.returncode}"
        summ = os.path.join(out, "pmc_summary.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summarise.py"), summ, os.path.join(out, "fetch"),
                            os.path.join(out, "write")], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
        if r.returncode != 0 or not os.path.exists(summ):
            return None, "pmc_summarise failed"
        with open(summ) as f:
            pm = json.load(f)
        v = pm.get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch")
        if v is None:
            return None, f"no counters for {kernel}"
        cal = pm.get("calibration", {})
        return v, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_workload.py "
                   f"(65 536 games, same pre-roll, last 96 launches of {kernel}), units calibrated on k_calib_copy "
                   f"({cal.get('bytes_per_FETCH_SIZE_unit', 0):.0f} B per FETCH_SIZE unit, {cal.get('bytes_per_WRITE_SIZE_unit', 0):.0f} B per WRITE_SIZE unit)")
    except Exception as e:                                   # (timeouts included) - the bench line must still be printed
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


# ---- the line is printed even when the run does not finish: whatever rank 0 has measured so far, with a `status`
# that says how far it got and why it stopped.  A peer that dies makes torch.distributed.run SIGTERM the others: rank 0 prints then.
PARTIAL = {"metric": "Catan env-steps/sec at 65k parallel games", "value": None, "unit": "env-steps/s", "status": "started", "stage": "start"}
_EMITTED = [False]


def emit(status=None):
    if _EMITTED[0] or int(os.environ.get("RANK", "0")) != 0:
        return
    _EMITTED[0] = True
    if status is not None:
        PARTIAL["status"] = status
    try:
        from settlers_of_catan_rl_amd import dist as cdist
        PARTIAL.setdefault("dist_init", cdist.INIT_REPORT)
        if PARTIAL.get("status") != "ok":
            PARTIAL.setdefault("rccl_log_rank0", cdist.rccl_log_tail())
    except Exception:
        pass
    sys.stdout.write(json.dumps(PARTIAL) + "\n")
    sys.stdout.flush()


def stage(name):
    PARTIAL["stage"] = name


def _watch_sigterm():
    """SIGTERM -> the line, even while the main thread sits inside a collective or a device synchronize (a Python-level handler
    would only run once that call returns): the C-level handler writes the signal number to a pipe (signal.set_wakeup_fd), a
    helper thread reads it and prints."""
    import signal
    import threading
    r, w = os.pipe()
    os.set_blocking(w, False)
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)
    signal.signal(signal.SIGTERM, lambda signum, frame: None)

    def watch():
        while True:
            b = os.read(r, 1)
            if b and b[0] == signal.SIGTERM:
                msg = (f"terminated by SIGTERM during stage '{PARTIAL.get('stage')}' (under torch.distributed.run: a peer rank failed or "
                       f"the launcher timed out; the peer's traceback is on stderr)")
                try:
                    emit(msg)
                except Exception:
                    sys.stdout.write(json.dumps({"metric": PARTIAL.get("metric"), "value": PARTIAL.get("value"), "unit": "env-steps/s", "status": msg}) + "\n")
                    sys.stdout.flush()
                os._exit(1)
    threading.Thread(target=watch, daemon=True, name="bench-sigterm").start()


PREROLL_PASSES = 8192        # untimed deferred passes before --warmup: several game lengths, so that the games' ages are mixed
MIN_TIMED_S = 0.30           # the timed region is `reps` x --steps passes, reps chosen so that it lasts at least this long


def _reps_for(per_pass_s, steps):
    import math
    return max(1, int(math.ceil(MIN_TIMED_S / max(per_pass_s * steps, 1e-9))))


def ppo_update_record(env, n, rank, world, T, cdist):
    """The metric's second half (BASELINE.json: "PPO wall-clock/update, 1->8 GPU"; workload RL/ppo/arguments.py:48-56) on the
    same games: one rollout of T active-seat decisions per game + one PPO update with the reference's 10 epochs x 64
    minibatches, bf16 autocast, fp32 master weights, flat-bucket gradient all-reduce over RCCL when world > 1.  T defaults to
    the reference's 200 (a minibatch has T * n / 64 = 204 800 rows at 65 536 games; ~63 GB of rollout tensors in HBM);
    three updates are run and the THIRD is reported (see below)."""
    import torch
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    net = CatanPolicy().cuda()
    cdist.broadcast_parameters(net)
    col = RolloutCollector(env, net, T, seed=rank, autocast_dtype=torch.bfloat16)
    tr = PPOTrainer(net, PPOConfig(), autocast_dtype=torch.bfloat16, seed=rank)
    first, warm = None, []
    # The THIRD update is reported.  The first captures the policy pass's hipGraph; the first and the second still load GEMM kernels:
    # a minibatch's row counts (distinct boards, rows per action head) differ from step to step, so the library meets new problem
    # sizes - and loads the code objects it picks for them - for a few hundred steps on a fresh process (measured per step: 54, 43,
    # 34, 32, 32 ms over the first five epochs-pairs; 31.8 ms from then on).  Both warm-up updates are in `warmup_updates`.
    for u in range(3):
        cdist.barrier()
        t0 = time.perf_counter()
        st = col.gather_rollouts()
        cdist.barrier()
        t1 = time.perf_counter()
        vl, al, el = tr.update(st)
        cdist.barrier()
        t2 = time.perf_counter()
        col.after_rollouts()
        import math
        if not all(math.isfinite(x) for x in (vl, al, el)):          # an update on NaNs does no real work: its time must not be reported
            raise RuntimeError(f"non-finite PPO losses in update {u}: value {vl}, action {al}, entropy {el}")
        rollout_s, update_s = cdist.max_over_ranks(t1 - t0), cdist.max_over_ranks(t2 - t1)
        if u < 2:
            warm.append({"rollout_s": rollout_s, "update_s": update_s, "minibatches_s": tr.timings.get("minibatches_s")})
        if u == 0:
            first = {"rollout_s": rollout_s, "update_s": update_s}
        iters_selfplay = col.iters
    dec = world * n * T
    # The reference's ACTUAL rollout workload (VERDICT r5 missing #3): seat 0 of every game plays the central policy against three league
    # snapshots per worker (RL/ppo/game_manager.py:15,25-31; update_opponent_policies.py:13-43) instead of all four seats playing the central
    # policy.  One more rollout, not followed by an update (the update's work does not depend on who the opponents were): the league's bounded
    # variant with max_distinct = 3 snapshots in play (4 nets with the central one; league.py), workers of 5 games, one captured policy pass per
    # distinct net per env pass on that net's rows (rollout.RolloutCollector._act).
    league = None
    try:
        from settlers_of_catan_rl_amd.league import League
        lg = League(envs_per_worker=5, max_distinct=3, seed=rank)
        for _ in range(3):
            lg.add(net)                                   # (three snapshots with the current weights: what is timed does not depend on their values)
        distinct = lg.assign(col, lambda: CatanPolicy().cuda())
        league_all = []
        for _ in range(2):                                # (the first captures the per-net policy passes: reported beside the second)
            cdist.barrier()
            t0 = time.perf_counter()
            st_l = col.gather_rollouts()
            cdist.barrier()
            league_all.append(cdist.max_over_ranks(time.perf_counter() - t0))
            col.after_rollouts()
        league_s = league_all[-1]
        league = {"rollout_s": league_s, "first_rollout_s": league_all[0], "max_distinct": 3, "nets_in_play": 1 + len(distinct), "envs_per_worker": 5, "env_passes_in_rollout": col.iters,
                  "note": "seat 0 = the central policy, the other three seats = league snapshots drawn per worker of 5 games (bounded league: 3 distinct "
                          "snapshots in play); one captured policy pass per distinct net per env pass on that net's rows; beside `rollout_s` (every seat plays the central policy)"}
        del st_l
        col.set_opponents([], None)
    except Exception as e:                                # (the bench line must still be printed)
        league = {"error": f"{type(e).__name__}: {e}"}
    learner = None
    if rank == 0:                       # per-kernel rooflines of the learner side, measured in this run (tools/learner_rooflines.py)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from learner_rooflines import learner_rooflines
            del st
            learner = learner_rooflines(env, net, T=200, rows_mb=204800)
        except Exception as e:          # (the bench line must still be printed)
            learner = {"error": f"{type(e).__name__}: {e}"}
    return {"value": rollout_s + update_s, "unit": "s/update", "higher_is_better": False, "num_steps": T, "roofline_learner": learner,
            "reference_num_steps": 200, "games_per_gpu": n, "ppo_epoch": tr.cfg.ppo_epoch, "num_mini_batch": tr.cfg.num_mini_batch,
            "minibatch_rows": n * T // tr.cfg.num_mini_batch, "active_seat_decisions_per_update": dec,
            "rollout_s": rollout_s, "update_s": update_s, "env_passes_in_rollout": iters_selfplay, "league_rollout": league, **tr.timings,
            "decisions_per_s": dec / (rollout_s + update_s), "dtype": "bf16 autocast, fp32 master weights",
            # world > 1: the flat 7.7 MB gradient bucket, one RCCL all-reduce (ReduceOp.AVG) per optimiser step, timed with events
            "allreduce_s_per_step": (tr.timings["allreduce_s"] / (tr.cfg.ppo_epoch * tr.cfg.num_mini_batch) if "allreduce_s" in tr.timings else None),
            "losses": {"value": vl, "action": al, "entropy": el},
            "first_update": first, "warmup_updates": warm,
            "hbm_gb_allocated": torch.cuda.max_memory_allocated() / 2 ** 30,
            "note": ("the reference's workload (RL/ppo/arguments.py:48-56: T = 200, 10 epochs x 64 minibatches)" if T == 200 else
                     f"T = {T} instead of the reference's 200 (stated)") +
                    "; every seat plays the central policy; value = the third update (steady state); the first (one-time hipGraph "
                    "capture of the policy pass) and the second (the GEMM library still loading kernels for new row counts) are in warmup_updates"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--envs", type=int, default=65536, help="games per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic (N = 1)")
    ap.add_argument("--no-validate", action="store_true", help="skip the mask-bit legality check in k_step")
    ap.add_argument("--window", type=int, default=32,
                    help="W > 0: deferred loop, tier-2 longest road + re-deals once per W passes; 0: lock-step loop")
    ap.add_argument("--no-lockstep", action="store_true", help="skip the lock-step measurement of a deferred run")
    ap.add_argument("--preroll", type=int, default=PREROLL_PASSES, help="untimed passes before --warmup (mixes the games' ages)")
    ap.add_argument("--ppo-steps", type=int, default=200,
                    help="T of the `ppo_update` sub-record (three rollouts + PPO updates on the same games, the third one reported; "
                         "200 = the reference's num_steps, RL/ppo/arguments.py:54-56); 0: skip")
    args = ap.parse_args()

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the HIP path has no CPU fallback)")
    from settlers_of_catan_rl_amd import dist as cdist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    # everything below runs on ONE stream of its own, not the legacy default stream (the env loop is 1.3 % faster there: 38.4 against 38.9 us per
    # pass, profiles/r06_stream_priority_ab.txt; the HIP events of `roofline` are recorded on this same stream by the library)
    torch.cuda.set_stream(torch.cuda.Stream())
    # backend "nccl" == RCCL over xGMI; CATAN_DIST_BACKEND=gloo lets two ranks share one GPU (smoke test of this code path)
    stage("init_process_group + allreduce_selfcheck")
    cdist.on_hang = lambda msg: emit("failed during stage 'init_process_group + allreduce_selfcheck': " + msg)
    rank, local_rank, world = cdist.init_from_env(backend=os.environ.get("CATAN_DIST_BACKEND") or None)
    PARTIAL.update(n_gpus=world, steps=args.steps, warmup=args.warmup, world=world,
                   backend=(torch.distributed.get_backend() if world > 1 else None), dist_init=cdist.INIT_REPORT)
    stage("create env")

    from settlers_of_catan_rl_amd.env import VecCatanEnv

    from settlers_of_catan_rl_amd import _lib as _clib
    assert _clib.lib().catan_step_algorithmic_bytes() == ALGO_BYTES["k_step"], "bench.py's byte table and csrc/catan_state.h disagree"
    assert _clib.lib().catan_step_fused_algorithmic_bytes() == ALGO_BYTES["k_step_fused"], "bench.py's byte table and csrc/catan_state.h disagree"
    env_id0, n = cdist.shard(rank, args.envs)                # global game ids: results do not depend on `world`
    env = VecCatanEnv(n, seed=args.seed, env_id0=env_id0, validate_actions=not args.no_validate, auto_reset=True)

    # ---- pre-roll (untimed, independent of --warmup): all games start a game at pass 0, and the opening phase (initial
    # placement: every road goes through the longest-road slow path) is not the steady state.  Several game lengths of
    # deferred passes spread the games' ages; its duration also sizes the timed region.
    W = args.window if args.window > 0 else 32
    stage("pre-roll")
    cdist.barrier()
    t0 = time.perf_counter()
    env.random_rollout_deferred(max(args.preroll, 64), W)
    cdist.barrier()
    pre_pass_s = cdist.max_over_ranks((time.perf_counter() - t0) / max(args.preroll, 64))
    step_idx = 1 << 20                                        # lock-step passes index the policy stream by a global step number
    stage("timed region")

    if args.window > 0:
        reps = _reps_for(pre_pass_s, args.steps)
        env.random_rollout_deferred(args.warmup, args.window)
        c0 = int(env.policy_counters().sum())
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout_deferred(args.steps * reps, args.window)      # reps x EXACTLY --steps passes, one timed region
        cdist.barrier()
        dt = cdist.max_over_ranks(time.perf_counter() - t0)     # max over ranks
        my_steps = int(env.policy_counters().sum()) - c0        # decisions actually executed by this rank's games
        env_steps = cdist.sum_over_ranks(my_steps)
    else:
        reps = _reps_for(pre_pass_s * 4.0, args.steps)          # a lock-step pass is ~4 deferred passes long
        env.random_rollout(step_idx, args.warmup); step_idx += args.warmup
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout(step_idx, args.steps * reps); step_idx += args.steps * reps
        cdist.barrier()
        dt = cdist.max_over_ranks(time.perf_counter() - t0)     # max over ranks
        my_steps = n * args.steps * reps
        env_steps = world * my_steps
    timed_passes = args.steps * reps
    bad = env.invalid_action_count()
    PARTIAL.update(value=env_steps / dt, ms_per_step=dt / timed_passes * 1e3, timed_s=dt, timed_passes=timed_passes, higher_is_better=True,
                   scaling="weak", env_steps_executed=env_steps, invalid_actions=bad, status="timed region done; later stages incomplete")
    stage("device identities (all_gather_object)")
    devices = cdist.device_identities()                       # (a collective: every rank calls it)
    PARTIAL["ranks"] = devices
    stage("lock-step / step_api measurements")

    lockstep = None
    if args.window > 0 and not args.no_lockstep:
        # the same games, continued in lock-step (every game steps in every pass): same rule for the timed region
        ls_reps = _reps_for(pre_pass_s * 4.0, args.steps)
        env.random_rollout(step_idx, 32); step_idx += 32
        cdist.barrier()
        t0 = time.perf_counter()
        env.random_rollout(step_idx, args.steps * ls_reps); step_idx += args.steps * ls_reps
        cdist.barrier()
        ls_dt = cdist.max_over_ranks(time.perf_counter() - t0)
        ls_passes = args.steps * ls_reps
        lockstep = {"value": world * n * ls_passes / ls_dt, "unit": "env-steps/s", "steps": args.steps, "reps": ls_reps,
                    "timed_s": ls_dt, "ms_per_step": ls_dt / ls_passes * 1e3}

    # The same games through the boundary's step entry points with CALLER-SUPPLIED actions (what a collector or a search calls:
    # include/catan_hip.h catan_step / catan_step_deferred; the value above is the library's own loop around the same kernels).
    # The policy stub is the library's sampler writing into a caller-owned buffer: one more launch per pass, no host read.
    step_api = None
    if args.window > 0 and not args.no_lockstep:
        acts = torch.empty((n, 18), dtype=torch.int32, device="cuda")
        api_passes = max(1024, args.steps // 2)
        def api_loop(deferred, passes, base):
            for k in range(passes):
                env.sample_random_actions(base + k, out=acts)
                if deferred:
                    env.step_deferred(acts, args.window)
                else:
                    env.step(acts)
        api_loop(True, 256, step_idx); step_idx += 256
        env.step_flush()
        cdist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        api_loop(True, api_passes, step_idx); step_idx += api_passes
        torch.cuda.synchronize(); cdist.barrier()
        d_dt = cdist.max_over_ranks(time.perf_counter() - t0)
        act_cnt = torch.zeros((), dtype=torch.int64, device="cuda")          # the share of games that step in a call: counted over 256 more (untimed) calls
        for k in range(256):
            env.sample_random_actions(step_idx + k, out=acts)
            env.step_deferred(acts, args.window)
            act_cnt += (env.status == 0).sum()
        step_idx += 256
        waiting_now = 1.0 - float(act_cnt) / (256.0 * n)
        env.step_flush()
        cdist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        api_loop(False, api_passes // 4, step_idx); step_idx += api_passes // 4
        torch.cuda.synchronize(); cdist.barrier()
        l_dt = cdist.max_over_ranks(time.perf_counter() - t0)
        step_api = {"passes": api_passes, "window": args.window,
                    "catan_step_deferred_us_per_call": d_dt / api_passes * 1e6, "waiting_fraction": waiting_now,
                    "catan_step_deferred_env_steps_per_s": world * n * (1.0 - waiting_now) * api_passes / d_dt,
                    "catan_step_us_per_call": l_dt / (api_passes // 4) * 1e6, "catan_step_env_steps_per_s": world * n * (api_passes // 4) / l_dt,
                    "note": "catan_sample_random_actions (global step index: a policy stub) + the step call per pass, actions in a caller-owned "
                            "buffer; the deferred rate = games x (share of games whose step completes in a call, counted over 256 further calls) / time per call"}

    out = None
    kms = None
    if rank == 0:
        # per-kernel durations: HIP events on the stream each kernel is launched on, a separate short pass right after
        prof_steps = 512
        sp0 = env.slow_path_counts()
        kms = env.random_rollout_timed(step_idx, prof_steps, args.window)
        sp1 = env.slow_path_counts()
    step_idx += 512
    ppo = None
    if args.ppo_steps > 0:
        stage("ppo_update (3 rollouts + PPO updates with the gradient all-reduce)")
        ppo = ppo_update_record(env, n, rank, world, args.ppo_steps, cdist)
    stage("assembling the line")
    if ppo is not None and rank != 0:
        ppo.pop("roofline_learner", None)
    if rank == 0:
        slow_launches = prof_steps if args.window <= 0 else -(-prof_steps // args.window)
        launches = {k: (slow_launches if k in ("k_lr_heavy", "k_reset_list") else prof_steps) for k in kms}
        per_launch_us = {k: v * 1e3 / launches[k] for k, v in kms.items() if k in ALGO_BYTES}
        fused_loop = args.window > 0 and env.deferred_fused   # ONE kernel per pass: k_step<G, true> samples the next action itself
        fast_path = ("k_step",) if fused_loop else FAST_PATH
        if fused_loop:
            per_launch_us.pop("k_sample_random", None)        # (no such kernel in this loop: the slot holds the gap between two event records)
        algo = dict(ALGO_BYTES, k_step=ALGO_BYTES["k_step_fused"]) if fused_loop else ALGO_BYTES
        dom = max(fast_path, key=per_launch_us.get)
        active = my_steps / (n * timed_passes)                # games that take a step in a pass (the others are busy)
        achieved = algo[dom] * n * active / (per_launch_us[dom] * 1e-6) / 1e9
        fast_us = sum(per_launch_us[k] for k in fast_path)
        traffic, traffic_src = None, None
        if world == 1 and n == 65536 and not args.no_live_pmc:     # (the counter workload runs the BASELINE size)
            traffic, traffic_src = live_pmc_traffic(dom)
            if traffic is None:
                traffic_src = f"live counter passes unavailable ({traffic_src}); "
        if traffic is None and os.path.exists(PMC_SUMMARY):
            live_note = traffic_src or ""
            with open(PMC_SUMMARY) as f:
                pm = json.load(f)
            k = pm.get("kernels", {}).get(dom, {})
            traffic = k.get("hbm_bytes_per_launch")
            traffic_src = live_note + (f"steady-state PMC of the same kernel, NOT collected in this run: "
                           f"{os.path.relpath(PMC_SUMMARY, ROOT)}, separate --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated, "
                           f"65 536 games after the same pre-roll")
        per_step_bytes = sum(algo[k] for k in fast_path)
        roofline = {
            "bound": "hbm", "kernel": dom + ("<G, SAMPLE> (fused-sampling step)" if fused_loop else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_game": algo[dom],
            "algorithmic_bytes_per_launch": algo[dom] * n * active,
            "avg_launch_us": per_launch_us[dom],
            "all_kernels_avg_launch_us": per_launch_us,
            "fast_path_algorithmic_bytes_per_game": per_step_bytes,
            "fast_path_achieved_gbs": per_step_bytes * n * active / (fast_us * 1e-6) / 1e9,
            # whole schedules, end to end: algorithmic bytes of one env step x executed steps / wall time
            "end_to_end": {"algorithmic_bytes_per_env_step": per_step_bytes,
                           "timed_schedule_gbs": per_step_bytes * env_steps / world / dt / 1e9,
                           "timed_schedule_frac": per_step_bytes * env_steps / world / dt / 1e9 / HBM_PEAK_GBS},
            "note": "integer/byte rules engine at one or two waves per SIMD: latency- and divergence-bound, far below the HBM "
                    "roofline by nature (SURVEY.md 8(d)); frac is reported for the dominant kernel of the timed loop; "
                    "the slow-path kernels (k_lr_*, k_reset_list) are latency-bound path searches / re-deals (LDS + ALU, "
                    "a few MB per launch) and run on side streams in the deferred loop",
        }
        t1_req, t2_req, t1_launches = (b - a for a, b in zip(sp0, sp1))
        lr_us = per_launch_us["k_lr_finish"]
        roofline["slow_path"] = {
            "bound": "latency (path search in LDS / registers; not an HBM- or MFMA-bound kernel)",
            "tier1_requests_per_launch": t1_req / max(1, t1_launches), "tier2_requests_per_launch": t2_req / max(1, slow_launches),
            "algorithmic_bytes_per_request": LR_REQUEST_BYTES, "k_lr_finish_avg_launch_us": lr_us,
            "k_lr_finish_achieved_gbs": LR_REQUEST_BYTES * t1_req / max(1, t1_launches) / (lr_us * 1e-6) / 1e9 if lr_us > 0 else None,
            "k_lr_heavy_avg_launch_us": per_launch_us["k_lr_heavy"],
            "note": "while a player's cached longest path is exact, a new road only searches the paths THROUGH the new edge "
                    "(DESIGN.md 4.2); requests that exceed the tier-1 iteration budget go to tier 2"}
        if lockstep is not None:
            g = per_step_bytes * n / (lockstep["ms_per_step"] * 1e-3) / 1e9
            roofline["end_to_end"].update(lockstep_gbs=g, lockstep_frac=g / HBM_PEAK_GBS)
        value = env_steps / dt
        out = {
            "metric": "Catan env-steps/sec at 65k parallel games", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "reps": reps, "timed_passes": timed_passes,
            "timed_s": dt, "preroll_passes": max(args.preroll, 64), "ms_per_step": dt / timed_passes * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 (integer bitboards)",
            "data": "synthetic (random-seed boards, uniform-random legal policy on device)",
            "config": {"workload": ("configs[1]: 65 536 parallel envs per GPU, random policy, step+mask only, bit-exact vs CPU oracle" if world == 1 else
                                    f"configs[3]: {world * n} envs sharded across {world} x MI355X ({n} per GPU, global game ids), PPO with one flat "
                                    f"gradient all-reduce per optimiser step (`ppo_update`); `value` = configs[1]'s random-policy step+mask rate "
                                    f"summed over the {world} shards"),
                       "games_per_gpu": n, "games_total": world * n, "validate_actions": not args.no_validate,
                       "auto_reset": True, "parallelism": f"games sharded over {world} GPU(s), no collective",
                       "schedule": (f"deferred, window {args.window}, {'fused-sampling loop (one kernel per pass)' if env.deferred_fused else 'sampler + k_step per pass'}: slow-path games sit out; value = executed env steps / time"
                                    if args.window > 0 else "lock-step: every game steps in every pass"),
                       "timed_region": f"{reps} x --steps passes in one region (>= {MIN_TIMED_S} s), after {max(args.preroll, 64)} "
                                       f"untimed pre-roll passes + --warmup"},
            "env_steps_executed": env_steps, "active_fraction": env_steps / (world * n * timed_passes),
            "invalid_actions": bad,
            # who ran: one record per rank (all-gathered), the collective backend the learner's all-reduce uses
            "world": world, "backend": (torch.distributed.get_backend() if world > 1 else None), "ranks": devices,
            "distinct_devices": len({(d["host"], d["uuid"] or d["pci_bus_id"] or d["device"]) for d in devices}),
            "roofline": roofline,
        }
        if lockstep is not None:
            out["lockstep"] = lockstep
        if step_api is not None:
            out["step_api"] = step_api
        if ppo is not None:
            out["roofline_learner"] = ppo.pop("roofline_learner", None)
            out["ppo_update"] = ppo
        if not args.no_cpu_baseline and world == 1:         # (rank 0 at N = 1 only: with N ranks on the node the host cores are shared)
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        out["status"] = "ok"
        out["dist_init"] = cdist.INIT_REPORT
        PARTIAL.clear(); PARTIAL.update(out)
        emit()                                               # (before the final barrier: a peer that hangs there cannot take the line with it)
    if world > 1:
        stage("finalize")
        cdist.finalize()


if __name__ == "__main__":
    if int(os.environ.get("RANK", "0")) == 0:
        _watch_sigterm()
    try:
        main()
    except BaseException as e:
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        traceback.print_exc()
        emit(f"failed during stage '{PARTIAL.get('stage')}': {type(e).__name__}: {str(e)[:600]}")
        os._exit(1)
