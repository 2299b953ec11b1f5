#!/bin/bash
# SQ counters of k_step (the evidence for "VALU-issue-bound"): three rocprofv3 --pmc passes over tools/pmc_workload.py
# (65 536 games after the pre-roll, deferred loop), per-launch means of the last launches of k_step.  Never combined with a trace domain.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05/k_step_pmc; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d $O/p1 -o pmc -- python $R/tools/pmc_workload.py > $O/p1.log 2>&1; echo "p1 rc=$?" >> $O/status.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -o pmc -- python $R/tools/pmc_workload.py > $O/p2.log 2>&1; echo "p2 rc=$?" >> $O/status.txt
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o pmc -- python $R/tools/pmc_workload.py > $O/p3.log 2>&1; echo "p3 rc=$?" >> $O/status.txt
python - $O <<'PY' > $O/../k_step_sq_counters.json
import csv, sys, collections, glob, json, os
O = sys.argv[1]
res = {}
for p in ("p1", "p2", "p3"):
    for f in glob.glob(os.path.join(O, p, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void catan::k_step") or "k_step<" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            tail = v[-96:]
            res[c] = {"per_launch_mean": sum(tail) / len(tail), "launches": len(tail)}
w = res.get("SQ_WAVES", {}).get("per_launch_mean")
out = {"kernel": "catan::k_step (the default games-per-wave instantiation)", "workload": "tools/pmc_workload.py: 65 536 games, deferred W = 32, the last 96 launches", "counters": res}
if w:
    d = {}
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"):
        if k in res: d[k + "_per_wave"] = res[k]["per_launch_mean"] / w
    wc = res.get("SQ_WAVE_CYCLES", {}).get("per_launch_mean")
    if wc:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in res: d[k + "_over_WAVE_CYCLES"] = res[k]["per_launch_mean"] / wc
        d["wave_cycles_per_wave"] = wc / w
    out["derived"] = d
print(json.dumps(out, indent=1))
PY
cat $O/status.txt; head -c 1500 $O/../k_step_sq_counters.json
