set -x
mkdir -p gpurun_out/s5
python -m pytest tests -m gpu -x -q > gpurun_out/s5/gpu_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/s5/status.txt
python bench.py > gpurun_out/s5/bench_default.json 2> gpurun_out/s5/bench_default.err; echo "bench rc=$?" >> gpurun_out/s5/status.txt
tail -3 gpurun_out/s5/gpu_tests.txt; cat gpurun_out/s5/bench_default.json | head -c 1500
