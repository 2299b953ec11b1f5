"""Workload for `rocprofv3 --kernel-trace --stats` (tools/profile_round6.sh): the kernels bench.py lists under `roofline_learner`,
launched by the same code (tools/learner_rooflines.py) at the same shapes, so that the AverageNs of every kernel in the stats file can
be set beside the HIP-event time of the bench line.  Prints the list as JSON."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from learner_rooflines import learner_rooflines
env = VecCatanEnv(65536, seed=0); env.random_rollout(0, 800)
net = CatanPolicy().cuda()
print(json.dumps(learner_rooflines(env, net)))
