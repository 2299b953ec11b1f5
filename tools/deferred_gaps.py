"""Diagnostics: from a rocprofv3 kernel trace of tools/pmc_workload.py-like deferred passes, the idle time between the main stream's
kernels (k_sample_random -> k_step -> next k_sample_random)."""
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(p)) if "k_step" in r["Kernel_Name"] or "k_sample_random" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-2000:]
g1 = g2 = d1 = d2 = n = 0
for a, b, c in zip(rows, rows[1:], rows[2:]):
    if "k_sample_random" in a["Kernel_Name"] and "k_step" in b["Kernel_Name"] and "k_sample_random" in c["Kernel_Name"]:
        g1 += int(b["Start_Timestamp"]) - int(a["End_Timestamp"]); g2 += int(c["Start_Timestamp"]) - int(b["End_Timestamp"])
        d1 += int(a["End_Timestamp"]) - int(a["Start_Timestamp"]); d2 += int(b["End_Timestamp"]) - int(b["Start_Timestamp"]); n += 1
print(f"{n} passes: k_sample_random {d1 / n / 1e3:.1f} us, gap {g1 / n / 1e3:.1f} us, k_step {d2 / n / 1e3:.1f} us, gap to the next pass {g2 / n / 1e3:.1f} us; pass {(d1 + d2 + g1 + g2) / n / 1e3:.1f} us")
