"""Summary of a rocprofv3 kernel trace of tools/trace_one_replay.py: the replay's kernels in start order with duration, grid and workgroup
size - which library GEMM is which layer (grid = output tiles), which kernels overlap on the forked streams."""
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_step" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
names_env = ("k_lr_", "k_reset", "k_install", "k_sample", "k_classify", "k_masks", "k_step")
win = [r for r in rows[a + 1:b] if not any(n in r["Kernel_Name"] for n in names_env)]
t0 = int(win[0]["Start_Timestamp"])
print("kernels", len(win), "window us", (max(int(r["End_Timestamp"]) for r in win) - t0) / 1e3)
cols = [c for c in rows[0].keys() if "Grid" in c or "Workgroup" in c or "Stream" in c or "Queue" in c]
print("columns:", cols)
for r in win:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= 15:
        print("%8.1f us at %8.1f  grid %s wg %s q %s  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, "x".join(r.get(c, "?") for c in cols if "Grid" in c),
                                                          "x".join(r.get(c, "?") for c in cols if "Workgroup" in c), r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))

print("---- all kernels of the replay by start time (us, start, queue, name)")
for r in win:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%7.1f %8.1f q%s %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:110]))
