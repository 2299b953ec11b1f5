import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import policy as P, nn_kernels, spec
from settlers_of_catan_rl_amd.forward_search import GraphedAct
B = 65536
torch.set_grad_enabled(False)          # as the collector runs it (RolloutCollector.gather_rollouts is @torch.no_grad)
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 800)
f, lists, lens = env.get_obs_rows(torch.bfloat16); masks = env.get_action_masks(); lens = lens.long()
net = P.CatanPolicy().cuda().inference_copy(torch.bfloat16)
nn_kernels.use_tuned_gemms()
def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:60s} gpu {a.elapsed_time(b) / n * 1e3:8.1f} us   wall {(time.perf_counter() - t0) / n * 1e6:8.1f} us", flush=True)
for forked in (True, False):
    for fused in (True, False):
        P._Branches.enabled = forked; nn_kernels.fused_heads_enabled = fused
        gen = torch.Generator(device="cuda").manual_seed(1)
        ga = GraphedAct(net, buckets=(B,), autocast_dtype=torch.bfloat16, generator=gen)
        ga(f, lists, lens, masks)
        st = ga.graphs[B]
        timeit(f"graph replay only   forked={forked} fused_heads={fused}", lambda: st["g"].replay())
        timeit(f"GraphedAct call     forked={forked} fused_heads={fused}", lambda: ga(f, lists, lens, masks, with_logp=True, clone=False))
