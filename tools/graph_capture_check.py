import sys, traceback
sys.path.insert(0, "/root/repo")
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy
from settlers_of_catan_rl_amd import forward_search as fs
env = VecCatanEnv(600, seed=0); env.random_rollout(0, 300)
f, lists, lens = env.get_obs(); masks = env.get_action_masks()
net = CatanPolicy().cuda().eval()
ga = fs.GraphedAct(net, autocast_dtype=torch.bfloat16)
try:
    st = ga._capture(512, f, lists, lens.long(), masks)
    print("capture ok")
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): v, a = ga(f[:300], lists[:300], lens[:300].long(), masks[:300])
    torch.cuda.synchronize(); print("graphed ms", (time.perf_counter() - t0) / 20 * 1e3, ga.failed)
    t0 = time.perf_counter()
    for _ in range(20):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16): net.act(f[:300], lists[:300], lens[:300].long(), masks[:300])
    torch.cuda.synchronize(); print("eager ms", (time.perf_counter() - t0) / 20 * 1e3)
except Exception:
    traceback.print_exc()
