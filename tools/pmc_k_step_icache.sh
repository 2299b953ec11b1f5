#!/bin/bash
# Instruction-cache counters of k_step (24 500 instructions = ~190 KB of code, every action type a different ~25 KB slice of it; the instruction
# cache is 64 KB per two CUs): rocprofv3 --pmc passes over tools/pmc_workload.py, per-launch means of the last launches.  Never combined with a trace domain.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05/k_step_icache; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/p1 -o pmc -- python $R/tools/pmc_workload.py > $O/p1.log 2>&1; echo "p1 rc=$?" >> $O/status.txt
timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p2 -o pmc -- python $R/tools/pmc_workload.py > $O/p2.log 2>&1; echo "p2 rc=$?" >> $O/status.txt
python - $O <<'PY' > $O/../k_step_icache_counters.json
import csv, sys, collections, glob, json, os
O = sys.argv[1]
res = {}
for p in ("p1", "p2"):
    for f in glob.glob(os.path.join(O, p, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            key = "k_step" if "k_step<" in kn else ("k_sample_random" if "k_sample_random" in kn else ("k_lr_finish" if "k_lr_finish" in kn else None))
            if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                tail = v[-96:]
                res.setdefault(k, {})[c] = sum(tail) / len(tail)
out = {"workload": "tools/pmc_workload.py: 65 536 games, the deferred loop, per-launch means over the last 96 launches", "counters": res, "derived": {}}
for k, d in res.items():
    dd = {}
    if d.get("SQC_ICACHE_REQ"): dd["icache_miss_ratio"] = d.get("SQC_ICACHE_MISSES", 0) / d["SQC_ICACHE_REQ"]
    if d.get("SQ_WAVES"):
        for c in ("SQC_ICACHE_REQ", "SQC_ICACHE_MISSES", "SQ_IFETCH"):
            if c in d: dd[c + "_per_wave"] = d[c] / d["SQ_WAVES"]
    if d.get("SQ_WAVE_CYCLES") and "SQ_IFETCH_LEVEL" in d: dd["SQ_IFETCH_LEVEL_over_WAVE_CYCLES"] = d["SQ_IFETCH_LEVEL"] / d["SQ_WAVE_CYCLES"]
    out["derived"][k] = dd
print(json.dumps(out, indent=1))
PY
cat $O/status.txt; cat $O/../k_step_icache_counters.json; tail -3 $O/p1.log $O/p2.log
