"""Summary of a rocprofv3 --kernel-trace of tools/pmc_workload.py (or any run of the deferred random-policy loop): the last PASSES of
catan_random_rollout_deferred as the device saw them - per kernel its duration, and on the MAIN stream (the chain sampler -> k_step, or
k_step alone with fused sampling) the idle time between one kernel's end and the next one's start.  A pass's period = durations + gaps.
    python tools/pass_gaps.py <rocprof output dir> [passes]"""
import csv, glob, sys
from collections import defaultdict
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void catan::", "").split("(")[0][:40]
steps = [i for i, r in enumerate(rows) if "k_step" in r["Kernel_Name"]]
a, b = steps[-N - 1], steps[-1]
win = rows[a:b + 1]
main = [r for r in win if "k_step" in r["Kernel_Name"] or "k_sample_random" in r["Kernel_Name"]]
period = (int(main[-1]["Start_Timestamp"]) - int(main[0]["Start_Timestamp"])) / N / 1e3
print(f"last {N} passes: period {period:.2f} us per pass")
dur = defaultdict(list); gap = defaultdict(list)
for r in win:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for x, y in zip(main[:-1], main[1:]):
    gap[short(x["Kernel_Name"]) + " -> " + short(y["Kernel_Name"])].append((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"  {k:42s} x{len(v) / N:5.2f} per pass   mean {sum(v) / len(v):7.2f} us   p50 {v[len(v) // 2]:7.2f}   max {v[-1]:7.2f}")
print("main-stream gaps (end of one kernel -> start of the next):")
for k, v in gap.items():
    v = sorted(v)
    print(f"  {k:70s} mean {sum(v) / len(v):6.2f} us   p50 {v[len(v) // 2]:6.2f}   max {v[-1]:6.2f}")
# what else runs meanwhile: the side streams' kernels overlapping k_step
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win if "k_step" in r["Kernel_Name"]]
side = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in win if not ("k_step" in r["Kernel_Name"] or "k_sample_random" in r["Kernel_Name"])]
ov = defaultdict(float)
for s, e, n in side:
    for a_, b_ in ks:
        o = min(e, b_) - max(s, a_)
        if o > 0: ov[n] += o / 1e3
print("side-stream kernel time that overlaps a k_step (us per pass):", {k: round(v / N, 2) for k, v in ov.items()})

# is the main stream WAITING for tier 1?  per sampler launch: its start minus the end of the latest k_lr_finish that ended before it
# (small and constant = the sampler starts as soon as that tier-1 launch is over); per k_lr_finish: its start minus the end of the k_step before it
import bisect
lf = sorted((int(r["End_Timestamp"]), int(r["Start_Timestamp"])) for r in win if "k_lr_finish" in r["Kernel_Name"])
lf_end = [e for e, _ in lf]
d1 = []
for r in win:
    if "k_sample_random" in r["Kernel_Name"]:
        s0 = int(r["Start_Timestamp"]); i = bisect.bisect_right(lf_end, s0) - 1
        if i >= 0: d1.append((s0 - lf_end[i]) / 1e3)
ks_end = sorted(e for _, e in ks)
d2 = []
for e, s0 in lf:
    i = bisect.bisect_right(ks_end, s0) - 1
    if i >= 0: d2.append((s0 - ks_end[i]) / 1e3)
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))] if v else float("nan")
print(f"sampler start - end of the latest finished k_lr_finish: p10 {q(d1, .1):.2f}  p50 {q(d1, .5):.2f}  p90 {q(d1, .9):.2f} us")
print(f"k_lr_finish start - end of the k_step before it:        p10 {q(d2, .1):.2f}  p50 {q(d2, .5):.2f}  p90 {q(d2, .9):.2f} us")
