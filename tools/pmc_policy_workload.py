"""Workload for the learner-side rocprofv3 passes (tools/profile_policy_round.sh): three PPO-style training steps of the
policy net at the config-3 minibatch width (204 800 rows, bf16 observations and bf16 autocast as the trainer runs it) after a warm-up, so that the kernel-trace statistics and the MFMA counters
describe the steady-state step."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy

B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, a, _ = net.act(f, lists, lens, masks)
MB = int(os.environ.get("ROWS", "204800"))
rep = -(-MB // B)
f, lists, lens, masks, a = (t.repeat((rep,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks, a))
f = f.to(torch.bfloat16)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
for _ in range(int(os.environ.get("STEPS", "4"))):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v, lp, ent = net.evaluate_actions(f, lists, lens, masks, a)
    loss = v.float().mean() + lp.float().mean() - 0.01 * ent
    opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print("policy workload done")
