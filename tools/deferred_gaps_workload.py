"""Workload for tools/deferred_gaps.py: `rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/deferred_gaps_workload.py` (REPO = the repo root), then `python tools/deferred_gaps.py DIR`."""
import os, sys
sys.path.insert(0, os.environ.get("REPO", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(3072, 32)
torch.cuda.synchronize()
env.random_rollout_deferred(1024, 32)
torch.cuda.synchronize()
