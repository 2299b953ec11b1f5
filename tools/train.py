#!/usr/bin/env python
"""Self-play PPO training on the batched HIP env - the loop of the reference's RL/robust_train.py with its defaults
(RL/ppo/arguments.py), one process per GPU (torch.distributed over RCCL when launched with torch.distributed.run).

    python tools/train.py --envs 65536 --updates 10 --league 8
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=640, help="games per GPU (reference: 128 processes x 5 envs)")
    ap.add_argument("--num-steps", type=int, default=200)
    ap.add_argument("--ppo-epoch", type=int, default=10)
    ap.add_argument("--num-mini-batch", type=int, default=64)
    ap.add_argument("--updates", type=int, default=2)
    ap.add_argument("--league", type=int, default=8, help="max distinct opponent snapshots in play (0: pure self-play)")
    ap.add_argument("--eval-every", type=int, default=25)
    ap.add_argument("--num-eval-episodes", type=int, default=128)
    ap.add_argument("--eval-max-steps", type=int, default=2500)
    ap.add_argument("--checkpoint", type=str, default="")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lstm", action="store_true", help="include_lstm (build_agent_model.py:26): LSTM policy + truncated BPTT")
    args = ap.parse_args()
    import torch
    from settlers_of_catan_rl_amd import dist as cdist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, local_rank, world = cdist.init_from_env()
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    from settlers_of_catan_rl_amd.train import PPOTrainer, PPOConfig
    from settlers_of_catan_rl_amd.league import League
    from settlers_of_catan_rl_amd import evaluation, train_loop
    torch.manual_seed(args.seed)
    env_id0, n = cdist.shard(rank, args.envs)
    env = VecCatanEnv(n, seed=args.seed, env_id0=env_id0)          # game_manager.py:16: EnvWrapper() defaults (sparse win reward)
    net = CatanPolicy(include_lstm=args.lstm).cuda()
    cdist.broadcast_parameters(net)
    col = RolloutCollector(env, net, args.num_steps, seed=rank, autocast_dtype=torch.bfloat16)
    tr = PPOTrainer(net, PPOConfig(ppo_epoch=args.ppo_epoch, num_mini_batch=args.num_mini_batch), seed=rank)
    random_net = CatanPolicy(include_lstm=args.lstm).cuda().eval()                                   # robust_train.py:76-78: the evaluation opponent

    def evaluate(policy, update_num):
        return evaluation.run_evaluation_protocol(lambda m: VecCatanEnv(m, seed=args.seed + 1000 + update_num, env_id0=1 << 40, auto_reset=False),
                                                  policy, random_net, args.num_eval_episodes, update_num,
                                                  max_steps=args.eval_max_steps, autocast_dtype=torch.bfloat16)

    lg = League(max_distinct=args.league, seed=rank) if args.league > 0 else None
    targs = train_loop.TrainArgs(num_steps=args.num_steps, eval_every=args.eval_every, num_eval_episodes=args.num_eval_episodes)
    loop = train_loop.TrainingLoop(env, net, col, tr, targs, league=lg, make_net=lambda: CatanPolicy(include_lstm=args.lstm).cuda(),
                                   evaluate=evaluate if rank == 0 else None, checkpoint_path=args.checkpoint or None)
    for _ in range(args.updates):
        out = loop.run_update()
        if rank == 0:
            if out["eval"]:
                print(out["eval"])
            print(json.dumps({k: v for k, v in out.items() if k != "eval"}), flush=True)
    cdist.finalize()


if __name__ == "__main__":
    main()
