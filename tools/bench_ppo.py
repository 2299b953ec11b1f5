#!/usr/bin/env python
"""Config 3 (BASELINE.json): 65 536 games of 4-seat self-play PPO on one MI355X (per rank) - wall-clock per update and
its split (rollout collection / value recompute / GAE / minibatch SGD).

    python tools/bench_ppo.py --envs 65536 --num-steps 200            # reference T (arguments.py:54-56); ~100 GB of HBM
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/bench_ppo.py --envs 65536                                  # config 4: 524 288 games, RCCL gradient all-reduce;
                                                                         # `split.allreduce_s` = device time inside the all-reduces

Self-play here = every seat plays the central policy (the reference's league opponents are a `next` row).
Prints one JSON line on rank 0.  Not the driver's bench (that is bench.py = config 2)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--num-steps", type=int, default=200)
    ap.add_argument("--ppo-epoch", type=int, default=10)
    ap.add_argument("--num-mini-batch", type=int, default=64)
    ap.add_argument("--updates", type=int, default=1)
    ap.add_argument("--warm-games", type=int, default=600, help="random-policy steps before the first rollout (mixes game ages)")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--lstm", action="store_true", help="include_lstm: LSTM(512 -> 256) policy, truncated-BPTT minibatches (seq len 10)")
    ap.add_argument("--league", type=int, default=0,
                    help="K > 0: opponents from the snapshot league, at most K distinct nets in play (league.League, bounded "
                         "variant); 0: every seat plays the central policy")
    args = ap.parse_args()
    import torch
    from settlers_of_catan_rl_amd import dist as cdist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, local_rank, world = cdist.init_from_env()
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    torch.manual_seed(0)
    env_id0, n = cdist.shard(rank, args.envs)
    env = VecCatanEnv(n, seed=0, env_id0=env_id0)
    env.random_rollout(0, args.warm_games)
    make_net = lambda: CatanPolicy(include_lstm=args.lstm).cuda()
    net = make_net()
    cdist.broadcast_parameters(net)                         # every rank starts from rank 0's weights
    ac = None if args.fp32 else torch.bfloat16
    col = RolloutCollector(env, net, args.num_steps, seed=rank, autocast_dtype=ac)
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=args.ppo_epoch, num_mini_batch=args.num_mini_batch), autocast_dtype=ac, seed=rank)
    lg = None
    if args.league > 0:
        from settlers_of_catan_rl_amd.league import League
        lg = League(max_distinct=args.league, seed=rank)
        lg.add(net)                                          # robust_train.py:62-64: the deque starts with the initial policy
        lg.assign(col, make_net)
    res = []
    for u in range(args.updates):
        cdist.barrier()
        t0 = time.perf_counter()
        st = col.gather_rollouts()
        cdist.barrier()
        t1 = time.perf_counter()
        vl, al, el = tr.update(st)
        cdist.barrier()
        t2 = time.perf_counter()
        col.after_rollouts()
        if lg is not None and lg.after_update(u, net):
            lg.assign(col, make_net)
        res.append(dict(rollout_s=cdist.max_over_ranks(t1 - t0), update_s=cdist.max_over_ranks(t2 - t1), env_iters=col.iters,
                        value_loss=vl, action_loss=al, entropy_loss=el, **tr.timings))
    if rank == 0:
        last = res[-1]
        dec = world * n * args.num_steps
        print(json.dumps({
            "metric": "PPO wall-clock per update", "value": last["rollout_s"] + last["update_s"], "unit": "s/update",
            "higher_is_better": False, "n_gpus": world, "dtype": "fp32" if args.fp32 else "bf16 autocast (fp32 params/softmax)",
            "config": {"workload": "configs[2]: self-play PPO, RL/models net", "games_per_gpu": n, "num_steps": args.num_steps,
                       "ppo_epoch": args.ppo_epoch, "num_mini_batch": args.num_mini_batch, "league_max_distinct": args.league,
                       "active_seat_decisions_per_update": dec, "minibatch_rows": dec // world // args.num_mini_batch},
            "split": last, "decisions_per_s": dec / (last["rollout_s"] + last["update_s"]),
            "env_steps_per_s_in_rollout": world * n * last["env_iters"] / last["rollout_s"],
            "invalid_actions": env.invalid_action_count(),
            "hbm_gb_allocated": torch.cuda.max_memory_allocated() / 2 ** 30, "all_updates": res}))
    cdist.finalize()


if __name__ == "__main__":
    main()
