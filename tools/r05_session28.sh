#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ppo_pipeline.py -q -m gpu -x -k "recurrent_given" ) > $O/gpu_tests_28.txt 2>&1; echo rc=$?; tail -5 $O/gpu_tests_28.txt
