"""Aggregates rocprofv3 counter-collection CSVs (one directory per --pmc pass) into profiles/rNN_pmc_summary.json:
mean counter value per launch over the LAST `TAIL` launches of every catan kernel (the measured section of
tools/pmc_workload.py: 96 deferred passes after the pre-roll), calibrated with k_calib_copy (known bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CALIB_BYTES = 256 << 20
TAIL = int(os.environ.get("PMC_TAIL", "96"))
out_path = sys.argv[1]
dirs = sys.argv[2:]
vals = defaultdict(list)                     # (kernel, counter) -> [(dispatch id, value)]
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "catan::" not in name:
                    continue
                short = name.split("catan::")[1].split("(")[0]
                if os.environ.get("PMC_KEEP_TEMPLATE_ARGS") != "1":
                    short = short.split("<")[0]
                vals[(short, row["Counter_Name"])].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
kern = defaultdict(dict)
for (k, c), lst in vals.items():
    lst.sort()
    tail = [v for _, v in lst[-TAIL:]]
    kern[k][c] = {"mean_per_launch": sum(tail) / len(tail), "launches_averaged": len(tail), "launches_profiled": len(lst)}
calib = kern.get("k_calib_copy", {})
res = {"calibration": {"kernel": "k_calib_copy", "bytes_read": CALIB_BYTES, "bytes_written": CALIB_BYTES}, "kernels": {}}
fr = fw = None
if "FETCH_SIZE" in calib:
    fr = CALIB_BYTES / calib["FETCH_SIZE"]["mean_per_launch"]
    res["calibration"]["FETCH_SIZE_raw"] = calib["FETCH_SIZE"]["mean_per_launch"]
    res["calibration"]["bytes_per_FETCH_SIZE_unit"] = fr
if "WRITE_SIZE" in calib:
    fw = CALIB_BYTES / calib["WRITE_SIZE"]["mean_per_launch"]
    res["calibration"]["WRITE_SIZE_raw"] = calib["WRITE_SIZE"]["mean_per_launch"]
    res["calibration"]["bytes_per_WRITE_SIZE_unit"] = fw
for k, cs in sorted(kern.items()):
    e = {c: v for c, v in cs.items()}
    if fr is not None and "FETCH_SIZE" in cs:
        e["hbm_read_bytes_per_launch"] = cs["FETCH_SIZE"]["mean_per_launch"] * fr
    if fw is not None and "WRITE_SIZE" in cs:
        e["hbm_written_bytes_per_launch"] = cs["WRITE_SIZE"]["mean_per_launch"] * fw
    if "hbm_read_bytes_per_launch" in e and "hbm_written_bytes_per_launch" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_written_bytes_per_launch"]
    res["kernels"][k] = e
with open(out_path, "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in res["kernels"].items()}))
