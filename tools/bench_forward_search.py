#!/usr/bin/env python
"""Config 5 (BASELINE.json): batched forward search - 4 096 root games x 64 look-ahead simulations on one MI355X.

Each root game's deciding player proposes <= 10 root actions and gets `--sims` simulations (state broadcast ->
randomise_uncertainty -> proposed action -> up to `--depth` further decisions of that player, the policy acting for every
seat) allocated by the reference's UCB rule in rounds of `--round` simulations per root; 4 096 x 16 = 65 536 simulation
games are in flight at a time.  Prints one JSON line: simulations/s, root decisions/s, split.  Random-init policy
weights (no checkpoints without a network); bf16 autocast."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--roots", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--round", type=int, default=16)
    ap.add_argument("--depth", type=int, default=15, help="max_depth of a simulation (SURVEY 8(d) config 5: 15; the reference class default is 20)")
    ap.add_argument("--decisions", type=int, default=2, help="timed planner decisions per root (after one warm-up)")
    ap.add_argument("--warm-games", type=int, default=500, help="random-policy steps before the roots are taken (config 5: step 500)")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="eager policy calls for small batches too (default: hipGraph replays)")
    args = ap.parse_args()
    import torch
    from settlers_of_catan_rl_amd.env import VecCatanEnv
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    from settlers_of_catan_rl_amd import forward_search as fs
    torch.manual_seed(0)
    root = VecCatanEnv(args.roots, seed=0)
    root.random_rollout(0, args.warm_games)
    net = CatanPolicy().cuda().eval()
    ac = None if args.fp32 else torch.bfloat16
    search = fs.ForwardSearch(net, lambda n: VecCatanEnv(n, seed=1, env_id0=1 << 32, dense_reward=True, auto_reset=False), args.roots,
                              max_depth=args.depth, sims_per_root=args.sims, sims_per_round=args.round, autocast_dtype=ac,
                              use_graphs=not args.no_graphs)
    times = []
    for d in range(args.decisions + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        chosen, info = search.act(root)
        a = torch.from_numpy(chosen).to(root.device).to(torch.int32)
        root.step(a)                                         # play the chosen moves: the next decision is a new position
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / args.decisions
    sims = args.roots * args.sims
    print(json.dumps({
        "metric": "forward-search simulations per second", "value": sims / dt, "unit": "simulations/s", "higher_is_better": True,
        "n_gpus": 1, "dtype": "fp32" if args.fp32 else "bf16 autocast",
        "config": {"workload": "configs[4]: forward_search_policy, batched", "roots": args.roots, "sims_per_root": args.sims,
                   "sims_in_flight": args.roots * args.round, "max_depth": args.depth, "max_init_actions": 10,
                   "small_batch_inference": "eager" if (args.no_graphs or search.graphed.failed) else "hipGraph replays (buckets %s)" % (search.graphed.buckets,)},
        "s_per_decision_batch": dt, "root_decisions_per_s": args.roots / dt,
        "mean_proposed_actions": float(info["n_proposed"].mean()), "invalid_actions_roots": root.invalid_action_count(),
        "invalid_actions_sims": search.sim_env.invalid_action_count(), "inconsistent_deals": search.sim_env.inconsistent_deal_count(),
        "hbm_gb_allocated": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
