"""Diagnostics: per-wave phase times of k_step inside the DEFERRED loop (fused sampling), last pass of a short rollout."""
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
waves = max(n // 16 + 17, 7128)                                # (the buffer is sized for 16 games per wave)
names = {0: "stage-in", 1: "validate+apply", 2: "request push", 6: "done/reward+masks", 7: "sample+append+write-back"}
acc = []
for rep in range(24):
    env.random_rollout_deferred(33 + rep, 32)
    out = np.zeros((waves, 8), dtype=np.uint32)
    L.catan_profile_read_waves(env.h, out.ctypes.data_as(C.c_void_p))
    acc.append(out[:n // 64 + 17].copy())      # k_step's waves (the rows behind them belong to the slow-path kernels)
L.catan_profile_enable(env.h, 0)
a = np.concatenate(acc).astype(np.float64)
a = a[a[:, 5] > 0]
for k, nm in names.items():
    print(f"{nm:26s} mean {a[:, k].mean() / 100:7.2f} us   p99 {np.percentile(a[:, k], 99) / 100:7.2f}   max {a[:, k].max() / 100:7.2f}")
v = a[:, 4].astype(np.int64)
sel = a[:, 5] != 10        # (roll waves use slot 4 for their own detail before the tail overwrites it: all waves end with the tail's value)
print(f"  of the tail: draw {(v & 0xFFFF).mean() / 100:6.2f} us (p99 {np.percentile(v & 0xFFFF, 99) / 100:6.2f}), append {(v >> 16).mean() / 100:6.2f} us (p99 {np.percentile(v >> 16, 99) / 100:6.2f})")
tot = a[:, [0, 1, 2, 6, 7]].sum(1)
print(f"{'sum per wave':26s} mean {tot.mean() / 100:7.2f} us   p99 {np.percentile(tot, 99) / 100:7.2f}   max {tot.max() / 100:7.2f}")
