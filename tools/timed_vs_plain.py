"""k_step's duration: HIP events of the instrumented loop (catan_random_rollout_timed) - run this under rocprofv3 --kernel-trace --stats with MODE=timed / plain to set AverageNs beside it"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
torch.cuda.set_stream(torch.cuda.Stream())
env = VecCatanEnv(65536, seed=0)
env.random_rollout_deferred(4096, 32)
torch.cuda.synchronize()
mode = os.environ.get("MODE", "timed")
if mode == "timed":
    k = env.random_rollout_timed(1 << 20, 2048, 32)
    print(json.dumps({"mode": mode, "k_step_us_by_events": round(k["k_step"] / 2048 * 1e3, 2), "k_lr_finish_us": round(k["k_lr_finish"] / 1024 * 1e3, 2)}))
else:
    env.random_rollout_deferred(2048, 32)
    torch.cuda.synchronize()
    print(json.dumps({"mode": mode}))
