#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/status10.txt
run() { name=$1; shift; ( "$@" ) > $O/$name.txt 2> $O/$name.err; echo "$name rc=$?" >> $O/status10.txt; }
run gpu_tests_10 timeout 900 python -m pytest tests/test_optim.py tests/test_gpu_wgrad_big.py -x -q -m gpu
cd /tmp
run ab_wgrad_big env SWITCHES=wgrad_big timeout 600 python $R/tools/ab_step_switches.py 16
run bench_10 timeout 700 python $R/bench.py --no-cpu-baseline --no-live-pmc
cat $O/status10.txt; tail -5 $O/gpu_tests_10.txt; cat $O/ab_wgrad_big.txt; tail -3 $O/ab_wgrad_big.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r05/bench_10.txt').read().strip().splitlines()[-1])
p=d.get('ppo_update') or {}; print(d.get('value'), d.get('status'), {k:p.get(k) for k in ('value','rollout_s','update_s','values_s','minibatches_s','losses','env_passes_in_rollout')})
PY
