"""Round-4's `tools/rollout_schedules.py` died with SIGSEGV at its FOURTH env + collector in one process (65 536 games, T = 200).
This is that loop with the evidence switched on: faulthandler (the Python stack of the faulting thread), device memory after every
set (torch's allocator and hipMemGetInfo), the garbage collector's view (what is still alive after `del`), and switches that
separate the suspects:

  MODE=plain        the tool's loop as it was (del + empty_cache)
  MODE=gc           + gc.collect() before empty_cache
  MODE=close        + env.close() (catan_destroy) explicitly, collector.close() where it exists
  SETS=6            how many env + collector + storage sets
"""
import faulthandler, gc, json, os, sys, time
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd.rollout import RolloutCollector, RolloutStorage

N = int(os.environ.get("GAMES", "65536")); T = int(os.environ.get("T", "200"))
MODE = os.environ.get("MODE", "plain"); SETS = int(os.environ.get("SETS", "6")); GATHERS = int(os.environ.get("GATHERS", "2"))


def mem(tag):
    free, total = torch.cuda.mem_get_info()
    alive = sum(1 for o in gc.get_objects() if isinstance(o, (RolloutCollector, RolloutStorage, VecCatanEnv)))
    print(json.dumps({"at": tag, "torch_allocated_GB": round(torch.cuda.memory_allocated() / 2**30, 2),
                      "torch_reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2), "device_used_GB": round((total - free) / 2**30, 2),
                      "collector_storage_env_objects_alive": alive, "gc_counts": gc.get_count()}), flush=True)


torch.manual_seed(0)
net = CatanPolicy().cuda()
mem("start")
for k in range(SETS):
    env = VecCatanEnv(N, seed=0)
    env.random_rollout(0, 600)
    col = RolloutCollector(env, net, T, seed=0, autocast_dtype=torch.bfloat16)
    mem(f"set {k}: built")
    secs = []
    for u in range(GATHERS):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = col.gather_rollouts()
        torch.cuda.synchronize(); secs.append(round(time.perf_counter() - t0, 3))
        col.after_rollouts()
    print(json.dumps({"set": k, "gather_s": secs, "iters": col.iters, "invalid": env.invalid_action_count()}), flush=True)
    if MODE == "close":
        if hasattr(col, "close"):
            col.close()
        env.close()
    del col, st, env
    if MODE in ("gc", "close"):
        gc.collect()
    torch.cuda.empty_cache()
    mem(f"set {k}: released")
print("completed", SETS, "sets", flush=True)
