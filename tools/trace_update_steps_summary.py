"""Summary of a rocprofv3 --kernel-trace of tools/trace_update_steps_workload.py: the last optimiser steps (delimited by k_ppo_loss),
wall time per step, device busy time (union of kernel intervals), idle time and where the idle gaps sit."""
import csv, glob, sys
from collections import defaultdict
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_ppo_loss" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
print("k_ppo_loss launches:", len(marks))
a, b = marks[-6], marks[-1]
steps = 5
win = rows[a:b]
t0 = int(win[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"steps {steps}: wall {(t1 - t0) / steps / 1e6:.3f} ms per step, kernels per step {len(win) / steps:.0f}, sum of durations {sum(dur(r) for r in win) / steps / 1e6:.3f} ms per step")
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in win)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
prev_name = ev[0][2]
for s, e, nm in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, prev_name, nm)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = nm
busy += cur_e - cur_s
idle = t1 - t0 - busy
print(f"device busy (union) {busy / steps / 1e6:.3f} ms per step, idle {idle / steps / 1e6:.3f} ms per step in {len(gaps) / steps:.0f} gaps (mean {sum(g for g, _, _ in gaps) / max(1, len(gaps)) / 1e3:.2f} us)")
hist = defaultdict(lambda: [0, 0])
for g, _, _ in gaps:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    hist[k][0] += g; hist[k][1] += 1
for k in ("<2us", "<5us", "<10us", "<50us", ">=50us"):
    print(f"  gaps {k:7s}: {hist[k][1] / steps:7.1f} per step, {hist[k][0] / steps / 1e3:8.1f} us per step")
big = sorted(gaps, key=lambda g: -g[0])[:12]
for g, pn, nn in big:
    print(f"  gap {g / 1e3:8.1f} us between {pn[:60]} -> {nn[:60]}")
agg = defaultdict(lambda: [0, 0])
for r in win:
    k = r["Kernel_Name"][:90]; agg[k][0] += dur(r); agg[k][1] += 1
print("---- kernels by total time (us per step, launches per step)")
for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:45]:
    print(f"{d / steps / 1e3:9.1f} us {c / steps:7.1f}  {k}")
small = [r for r in win if dur(r) < 8000]
print(f"launches under 8 us: {len(small) / steps:.0f} per step, {sum(dur(r) for r in small) / steps / 1e3:.1f} us per step")
