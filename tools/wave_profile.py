"""Diagnostics: contention-free per-wave phase times of k_step (every wave stores its own durations), by action type."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import _lib

n = 65536
env = VecCatanEnv(n, seed=0)
L = _lib.lib()
env.random_rollout_deferred(3000, 32)
L.catan_profile_enable(env.h, 2)
waves = max(n // 16 + 17, 7128)                                # the buffer is sized for 16 games per wave (+ one partial wave per action-type bin)
names = {0: "stage-in", 1: "validate+apply", 2: "request push", 6: "done/reward+masks", 7: "write-back"}
tn = ["no-op", "settle", "road", "city", "buy_dev", "play_knight", "exchange", "propose", "respond", "robber", "roll", "end_turn", "steal", "discard",
      "play_vp", "play_yop", "play_rb", "play_mono"]
acc = []
for step in range(48):
    env.random_rollout(100000 + step, 1)
    out = np.zeros((waves, 8), dtype=np.uint32)
    L.catan_profile_read_waves(env.h, out.ctypes.data_as(C.c_void_p))
    acc.append(out.copy())
L.catan_profile_enable(env.h, 0)
a = np.concatenate(acc).astype(np.float64)
a = a[a[:, 5] > 0]                                   # waves that did work
for k, nm in names.items():
    print(f"{nm:20s} mean {a[:, k].mean() / 100:7.2f} us   p99 {np.percentile(a[:, k], 99) / 100:7.2f}   max {a[:, k].max() / 100:7.2f}")
tot = a[:, [0, 1, 2, 6, 7]].sum(1)
print(f"{'sum per wave':20s} mean {tot.mean() / 100:7.2f} us   p99 {np.percentile(tot, 99) / 100:7.2f}   max {tot.max() / 100:7.2f}")
for t in range(1, 18):
    sel = a[a[:, 5] == t + 0]
    if len(sel):
        print(f"  {tn[t]:11s} waves {len(sel):6d}: apply {sel[:, 1].mean() / 100:6.2f}  masks {sel[:, 6].mean() / 100:6.2f}  total {sel[:, [0, 1, 2, 6, 7]].sum(1).mean() / 100:6.2f} (max {sel[:, [0, 1, 2, 6, 7]].sum(1).max() / 100:6.2f})")
for t in (10, 11, 12, 1, 8, 17, 15):                  # slot 3: validate (before the switch) | the switch itself << 16
    sel = a[a[:, 5] == t]
    if len(sel):
        v = sel[:, 3].astype(np.int64)
        print(f"  {tn[t]:11s} before the switch {(v & 0xFFFF).mean() / 100:5.2f} us, switch {(v >> 16).mean() / 100:5.2f} us")
sel = a[a[:, 5] == 10]                                # roll, slot 4: dice draws | tile scan + bank << 10 | hands + estimates << 20
if len(sel):
    v = sel[:, 4].astype(np.int64)
    print(f"  roll detail: dice {(v & 1023).mean() / 100:5.2f} us, tile scan + bank {((v >> 10) & 1023).mean() / 100:5.2f} us, "
          f"hands + estimates {((v >> 20) & 1023).mean() / 100:5.2f} us")
