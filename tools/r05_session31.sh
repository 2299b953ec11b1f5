#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29523 $R/bench.py --gpus 8 --steps 20 --warmup 5 --no-live-pmc --ppo-steps 0 ) > $O/bench_8rank_env.txt 2> $O/bench_8rank_env.err; echo rc=$?
python - <<'PY'
import json
try:
    d=json.loads(open('/root/repo/gpurun_out/r05/bench_8rank_env.txt').read().strip().splitlines()[-1])
    print(d.get('status'), d.get('value'), d.get('n_gpus'), d.get('backend'), d.get('distinct_devices'), d['config'].get('workload')[:80], d['dist_init'].get('preflight',{}).get('one_distinct_device_per_rank'))
except Exception as e:
    print('no line', e)
PY
grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*" $O/bench_8rank_env.err | tail -6 | cut -c1-300
