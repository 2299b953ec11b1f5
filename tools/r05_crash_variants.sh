#!/bin/bash
# Round 5: the SIGSEGV of tools/rollout_schedules.py at its fourth env + collector (hipGraphLaunch inside libamdhip64), variants that
# separate the suspects.  Run through gpurun from the repo root; every step's stdout + stderr + exit code land in gpurun_out/r05.
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P="LD_PRELOAD=$R/tools/native/libsegvbt.so"
run() { name=$1; shift; env "$@" > $O/$name.txt 2>&1; echo "rc=$?" >> $O/$name.txt; }
run perf_branches CONFIGS=3,3 timeout 300 python $R/tools/rollout_schedules.py
run perf_nobranches CONFIGS=3,3 CATAN_NO_BRANCHES=1 timeout 300 python $R/tools/rollout_schedules.py
run crash_nobranches ASWAS=1 GATHERS=1 CATAN_NO_BRANCHES=1 $P timeout 400 python -X faulthandler $R/tools/rollout_schedules.py
run crash_dynq0 ASWAS=1 GATHERS=1 DEBUG_HIP_DYNAMIC_QUEUES=0 $P timeout 400 python -X faulthandler $R/tools/rollout_schedules.py
run crash_fgq1 ASWAS=1 GATHERS=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 $P timeout 400 python -X faulthandler $R/tools/rollout_schedules.py
for f in perf_branches perf_nobranches crash_nobranches crash_dynq0 crash_fgq1; do echo "== $f"; grep -v "^python\|^/usr\|^/lib\|amdgpu.ids" $O/$f.txt | cut -c1-200 | tail -9; done
