"""Diagnostics: per-request phase times of k_lr_finish in lock-step steps (every one-game wave stores its own durations in
rows 4128.. of the per-wave profile buffer): record load, path search, holder logic, done / rewards, masks, write-back."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import _lib

n = 65536
env = VecCatanEnv(n, seed=0)
L = _lib.lib()
env.random_rollout_deferred(3000, 32)
L.catan_profile_enable(env.h, 2)
rows = max(n // 16 + 17, 7128)
names = {0: "record load", 1: "plan + search + cache", 2: "call", 5: "holder update", 3: "cut case", 4: "done / rewards", 6: "masks", 7: "write-back"}
acc = []
heavy = []
for step in range(48):
    L.catan_profile_enable(env.h, 2)
    env.random_rollout(100000 + step, 1)
    out = np.zeros((rows, 8), dtype=np.uint32)
    L.catan_profile_read_waves(env.h, out.ctypes.data_as(C.c_void_p))
    acc.append(out[4128:4128 + 2000].copy())
    h = out[6128:6128 + 1000].astype(np.int64)
    h = h[h[:, 1] > 0]
    heavy.append(h)
L.catan_profile_enable(env.h, 0)
a = np.concatenate(acc).astype(np.float64)
a = a[a[:, 7] > 0]
print(f"{len(a) / 48:.0f} completed requests per step")
for k, nm in names.items():
    print(f"{nm:24s} mean {a[:, k].mean() / 100:7.2f} us   p50 {np.percentile(a[:, k], 50) / 100:7.2f}   p99 {np.percentile(a[:, k], 99) / 100:7.2f}   max {a[:, k].max() / 100:7.2f}")
tot = a[:, list(names)].sum(1)
print(f"{'sum per request':24s} mean {tot.mean() / 100:7.2f} us   p50 {np.percentile(tot, 50) / 100:7.2f}   p99 {np.percentile(tot, 99) / 100:7.2f}   max {tot.max() / 100:7.2f}")

print("tier 2 (k_lr_heavy), per lock-step step: workgroups with a request, split, rounds, search us (max over workgroups), span of the launch us")
for h in heavy[:48]:
    if not len(h):
        print("  no tier-2 request"); continue
    t0 = h[:, 6].min()
    span = ((h[:, 7] - t0) & 0xFFFFFFFF).max() / 100
    thr = (h[:, 5] >> 8) & 1
    print(f"  wgs {len(h):3d} split {h[0, 5] & 255} through-parts {thr.sum():3d} rounds max {h[:, 1].max():3d} mean {h[:, 1].mean():5.1f}  set-up {h[:, 0].max() / 100:5.1f}  "
          f"search max {h[:, 2].max() / 100:6.1f} mean {h[:, 2].mean() / 100:6.1f}  arrive {h[:, 3].max() / 100:5.1f}  completion {h[:, 4].max() / 100:5.1f}  span {span:6.1f}")
