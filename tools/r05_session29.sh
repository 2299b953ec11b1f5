#!/bin/bash
# the N = 2 start-up chain live on a ONE-GPU box: RCCL bound to the device / RCCL lazy must fail (two ranks, one device) and the ranks must agree on gloo
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 20 --warmup 5 --ppo-steps 0 --no-live-pmc ) > $O/bench_2rank_one_gpu.txt 2> $O/bench_2rank_one_gpu.err; echo rc=$?
tail -c 1500 $O/bench_2rank_one_gpu.txt; echo; grep -v "amdgpu.ids" $O/bench_2rank_one_gpu.err | tail -12 | cut -c1-220
