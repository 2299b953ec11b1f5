#!/bin/bash
# Learner-side profiles on the MI355X box: kernel-trace statistics of the policy training step, then the MFMA-busy counter in
# its own pass (no trace domains with --pmc).  Outputs: gpurun_out/prof_policy_r02.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_policy_r02
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o pol -- python $REPO/tools/pmc_policy_workload.py > $OUT/stats.log 2>&1
STEPS=2 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pol -- python $REPO/tools/pmc_policy_workload.py > $OUT/pmc.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python - <<PY
import csv, glob, json, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for p in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"][:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0); mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui > 0:
        rows.append({"kernel": k, "launches": cnt[k], "gui_active_cycles": gui, "mfma_busy_cycles_sum_over_simds": mf,
                     "mfma_busy_per_simd_over_active": mf / (gui * 1024.0)})
rows.sort(key=lambda r: -r["gui_active_cycles"])
json.dump({"note": "SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs / (GRBM_GUI_ACTIVE x 1024) = fraction of the kernel's active time "
                   "an average SIMD's MFMA pipe was busy", "kernels": rows[:40]}, open("$OUT/mfma_summary.json", "w"), indent=1)
print(len(rows), "kernels")
PY
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls $OUT; tail -2 $OUT/stats.log; tail -2 $OUT/pmc.log; head -c 1500 $OUT/mfma_summary.json
