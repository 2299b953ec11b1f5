#!/bin/bash
# packed-mask categorical heads: parity + step timing with the switch on / off
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_ppo_pipeline.py tests/test_gpu_policy_fixture.py -m gpu -x -q > gpurun_out/r05/s34_tests.txt 2>&1
tail -5 gpurun_out/r05/s34_tests.txt
SWITCHES=cat_bits timeout 900 python tools/ab_step_switches.py 12 > gpurun_out/r05/s34_ab.txt 2>&1
tail -12 gpurun_out/r05/s34_ab.txt
