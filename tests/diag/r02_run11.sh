#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02_gpu_tests11.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r02_gpu_tests11.log
timeout -k 10 300 python bench.py --shard candidates --steps 2 --warmup 1 > gpurun_out/r02_bench_cand.json 2> gpurun_out/r02_bench_cand.log
echo "bench cand rc $?"; tail -2 gpurun_out/r02_bench_cand.log; cat gpurun_out/r02_bench_cand.json
timeout -k 10 120 python tests/diag/ar_ab.py 3 2>/dev/null
