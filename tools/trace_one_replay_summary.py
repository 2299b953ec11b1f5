import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_step" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
win = [r for r in rows[a + 1:b] if "k_" not in r["Kernel_Name"][:12] or True]
# drop the env's own kernels right after the first marker and before the second
names_env = ("k_lr_", "k_reset", "k_install", "k_sample", "k_classify", "k_masks")
win = [r for r in win if not any(n in r["Kernel_Name"] for n in names_env)]
t0 = int(win[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in win)
print("kernels in the replay:", len(win), "window us:", (t1 - t0) / 1e3, "sum of durations us:", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win) / 1e3)
# union busy time and gaps
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
busy = 0; cur_s, cur_e = ev[0]
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU busy (union) us:", busy / 1e3, "idle us:", (t1 - t0 - busy) / 1e3, "gaps:", len(gaps), "mean gap us:", sum(g for g, _ in gaps) / max(1, len(gaps)) / 1e3)
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for r in win:
    k = r["Kernel_Name"][:70]; agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:25]:
    print(f"{d / 1e3:8.1f} us {c:4d}  {k}")
# timeline in 20 buckets: busy fraction
