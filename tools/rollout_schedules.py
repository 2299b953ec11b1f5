"""Config 3's rollout (65 536 games, T = 200, bf16 policy) under the collector's schedules: all games in every policy pass vs the
captured buckets (only the games that still miss observations), catan_step vs catan_step_deferred (window W).  Seconds per
gather_rollouts (the second and third of three: the first captures the policy graphs), env iterations, share of game-iterations
that produced a stored decision."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector

if os.environ.get("CATAN_BRANCHES_IN_GRAPHS"):  # the policy pass captured WITH its forked streams (round 4's form: see policy._Branches.in_graphs)
    from settlers_of_catan_rl_amd import policy as _pol
    _pol._Branches._in_graphs = True
N = int(os.environ.get("GAMES", "65536")); T = int(os.environ.get("T", "200"))
torch.manual_seed(0)
net = CatanPolicy().cuda()
H = tuple(N >> k for k in range(5))
configs = [("all games, catan_step", dict(act_buckets=(N,), deferred_window=0)), ("halving buckets, catan_step", dict(act_buckets=H, deferred_window=0)),
           ("buckets, catan_step", dict(deferred_window=0)),
           ("buckets, deferred W=4", dict(deferred_window=4)), ("buckets, deferred W=8", dict(deferred_window=8)),
           ("buckets, deferred W=2", dict(deferred_window=2))]
if os.environ.get("ASWAS"):      # round 4's keyword arguments (its first three lines ran the default deferred_window = 4 under a "catan_step" label)
    configs = [("all games, W=4 (r4 label: catan_step)", dict(act_buckets=(N,))), ("halving buckets, W=4 (r4 label: catan_step)", dict(act_buckets=H)),
               ("buckets, W=4 (r4 label: catan_step)", dict())] + configs[3:]
if os.environ.get("CONFIGS"):    # e.g. CONFIGS=2,2,2,3: which of them, in which order
    configs = [configs[int(i)] for i in os.environ["CONFIGS"].split(",")]
GATHERS = int(os.environ.get("GATHERS", "3"))
only = os.environ.get("ONLY")
for name, kw in configs:
    if only and only not in name:
        continue
    env = VecCatanEnv(N, seed=0)
    env.random_rollout(0, 600)
    col = RolloutCollector(env, net, T, seed=0, autocast_dtype=torch.bfloat16, **kw)
    out = []
    for u in range(GATHERS):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = col.gather_rollouts()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append((round(dt, 3), col.iters))
        col.after_rollouts()
    print(json.dumps({"schedule": name, "gather_s": [o[0] for o in out], "iterations": [o[1] for o in out],
                      "stored_share": round(N * T / (N * out[-1][1]), 4), "invalid": env.invalid_action_count(),
                      "bucket_changes_last": getattr(col, "bucket_log", None)}), flush=True)
    del col, st, env
    torch.cuda.empty_cache()
