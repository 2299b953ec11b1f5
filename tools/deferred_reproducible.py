"""Diagnostics (GPU box): is a deferred rollout reproducible?  The same environment, seed and number of passes in this process
twice and (argv[1] = a file to save to / compare with) across processes: final records compared word for word."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv

n, W, iters = 65536, 32, 3072
outs = []
for rep in range(2):
    env = VecCatanEnv(n, seed=0)
    if os.environ.get("LR_BUDGET"):
        env.set_lr_budgets(16, int(os.environ["LR_BUDGET"]))
    env.random_rollout_deferred(iters, W)
    torch.cuda.synchronize()
    outs.append((env.export_state().cpu(), env.policy_counters().cpu()))
    print("   slow-path counts (tier-1 requests, handed to tier 2, launches):", env.slow_path_counts())
    del env
same = torch.equal(outs[0][0], outs[1][0])
print("two runs in one process: identical records", same, "- games that differ:", int((outs[0][0] != outs[1][0]).any(1).sum()),
      "- games whose decision counts differ:", int((outs[0][1] != outs[1][1]).sum()))
d = (outs[0][1].long() - outs[1][1].long())
print("decision-count differences (run 0 - run 1):", sorted(d[d != 0].tolist()))
if len(sys.argv) > 1:
    if os.path.exists(sys.argv[1]):
        ref = torch.load(sys.argv[1])
        print("against the other process: games that differ:", int((ref[0] != outs[0][0]).any(1).sum()), "- decision counts differ:", int((ref[1] != outs[0][1]).sum()))
    else:
        torch.save(outs[0], sys.argv[1])
