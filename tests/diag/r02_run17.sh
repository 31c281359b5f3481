#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_rng.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q > gpurun_out/r02_gpu_tests17.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r02_gpu_tests17.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r02_gpu_tests17.log | head -30; exit 1; }
timeout -k 10 120 python tests/diag/ar_ab.py 5 2>/dev/null
