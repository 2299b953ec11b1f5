"""Diagnostics: from a rocprofv3 kernel trace of tools/deferred_gaps_workload.py, the timeline of the fused-sampling deferred loop:
k_step durations and the idle time between consecutive k_steps on the main stream, split by whether a k_lr_finish launch overlaps."""
import csv, glob, sys
import numpy as np
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
ks = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_step" in r["Kernel_Name"]])[-1000:]
lf = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_lr_finish" in r["Kernel_Name"]])
hv = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_lr_heavy" in r["Kernel_Name"]])
dur = np.array([b - a for a, b in ks]) / 1e3
gap = np.array([ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]) / 1e3
def overlap(a, b, iv):
    return sum(max(0, min(b, y) - max(a, x)) for x, y in iv if y > a and x < b)
ov = np.array([overlap(a, b, lf) for a, b in ks]) / 1e3
ovh = np.array([overlap(a, b, hv) for a, b in ks]) / 1e3
print(f"{len(ks)} k_step launches: duration mean {dur.mean():.1f} us (p10 {np.percentile(dur, 10):.1f}, p50 {np.percentile(dur, 50):.1f}, p90 {np.percentile(dur, 90):.1f}), "
      f"gap to the next mean {gap.mean():.1f} us (p50 {np.percentile(gap, 50):.1f}, p90 {np.percentile(gap, 90):.1f}); pass {dur.mean() + gap.mean():.1f} us")
for name, sel in (("no k_lr_finish overlap", ov < 1), ("k_lr_finish overlaps >= 10 us", ov >= 10)):
    if sel.any():
        print(f"  {name}: {int(sel.sum())} launches, duration {dur[sel].mean():.1f} us")
print(f"  k_lr_heavy overlaps: {int((ovh > 1).sum())} launches, duration {dur[ovh > 1].mean() if (ovh > 1).any() else 0:.1f} us; without: {dur[ovh <= 1].mean():.1f} us")
lfd = np.array([b - a for a, b in lf[-500:]]) / 1e3
print(f"k_lr_finish: {len(lf)} launches, duration mean {lfd.mean():.1f} us (p90 {np.percentile(lfd, 90):.1f})")
