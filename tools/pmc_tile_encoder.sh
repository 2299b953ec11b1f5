cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/te_pmc; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/sq_names.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o pmc -- python $R/tools/pmc_tile_encoder_workload.py > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/p2 -o pmc -- python $R/tools/pmc_tile_encoder_workload.py > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o pmc -- python $R/tools/pmc_tile_encoder_workload.py > $O/p3.log 2>&1
find $O -name "*counter_collection.csv" | head
for p in p1 p2 p3; do f=$(find $O/$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "tile_encoder" in k:
        for c, v in acc[k].items(): print(f"{k} {c} per launch {v / n[(k, c)]:.4g}  (launches {n[(k, c)]})")
PY
done
tail -n 3 $O/p1.log
