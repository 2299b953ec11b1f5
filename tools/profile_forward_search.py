import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import forward_search as fs
torch.manual_seed(0)
R = 4096
root = VecCatanEnv(R, seed=0); root.random_rollout(0, 700)
net = CatanPolicy().cuda().eval()
def T(): torch.cuda.synchronize(); return time.perf_counter()
f, lists, lens = root.get_obs(); masks = root.get_action_masks()
t0 = T()
props, counts = fs.propose_actions(net, f, lists, lens, masks, 10, autocast_dtype=torch.bfloat16)
t1 = T(); print("propose", t1 - t0, "mean proposals", counts.mean())
sim = VecCatanEnv(R * 16, seed=1, env_id0=1 << 32, dense_reward=True, auto_reset=False)
blobs = root.export_state().repeat_interleave(16, dim=0)
ctrl = root.deciding_player().long().repeat_interleave(16)
t0 = T(); sim.import_state(blobs); t1 = T(); sim.randomise_uncertainty(ctrl); t2 = T()
print("import", t1 - t0, "randomise", t2 - t1)
init = torch.from_numpy(props[:, 0]).cuda().repeat_interleave(16, dim=0)
# instrument simulate: count passes
orig_act = net.act
calls = []
def act(*a, **kw):
    t = T(); out = orig_act(*a, **kw); calls.append((a[0].shape[0], T() - t)); return out
net.act = act
t0 = T(); v = fs.simulate(sim, net, ctrl, init, 20, autocast_dtype=torch.bfloat16); t1 = T()
print("simulate", t1 - t0, "passes", len(calls), "policy time", sum(c[1] for c in calls), "rows/pass first/last", calls[0][0], calls[-1][0], "mean rows", np.mean([c[0] for c in calls]))
