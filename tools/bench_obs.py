import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import spec
n = 65536
env = VecCatanEnv(n, seed=0); env.random_rollout(0, 800)
T = 8
st_f = torch.zeros((T, n, spec.OBS_FLOATS), dtype=torch.bfloat16, device="cuda")
st_l = torch.zeros((T, n, 5, 25), dtype=torch.int8, device="cuda"); st_n = torch.zeros((T, n, 5), dtype=torch.int8, device="cuda")
t = torch.randint(0, T, (n,), device="cuda"); sel = torch.rand(n, device="cuda") < 0.25
def timeit(name, fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:60s} {a.elapsed_time(b) / reps * 1e3:8.1f} us", flush=True)
out32 = env.get_obs(); out16 = env.get_obs_rows(torch.bfloat16)
timeit("catan_obs fp32 dense", lambda: env.get_obs(out32))
timeit("catan_obs_rows bf16 dense", lambda: env.get_obs_rows(torch.bfloat16, out=out16))
timeit("catan_obs_rows bf16 dense + 25% storage rows", lambda: env.get_obs_rows(torch.bfloat16, out=out16, rows=(st_f, st_l, st_n), t=t, sel=sel))
timeit("catan_obs_rows fp32 dense", lambda: env.get_obs_rows(torch.float32, out=out32))
