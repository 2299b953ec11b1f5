#!/bin/bash
# the whole N = 2 line (env + ppo_update with the flat-bucket all-reduce, GAE statistics over ranks) on ONE GPU over gloo: the learner's N > 1 path at full size
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
( timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 2 --steps 20 --warmup 5 --no-live-pmc ) > $O/bench_2rank_full.txt 2> $O/bench_2rank_full.err; echo rc=$?
python - <<'PY'
import json
try:
    d=json.loads(open('/root/repo/gpurun_out/r05/bench_2rank_full.txt').read().strip().splitlines()[-1])
    p=d.get('ppo_update') or {}
    print(d.get('status'), d.get('value'), d.get('backend'), {k:p.get(k) for k in ('value','rollout_s','values_s','minibatches_s','allreduce_s_per_step','losses')})
except Exception as e:
    print('no line', e)
PY
grep -v "amdgpu.ids\|socket.cpp" $O/bench_2rank_full.err | tail -8 | cut -c1-300
