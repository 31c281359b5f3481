#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_bench_shapes.py tests/test_gpu_e2e.py -m gpu -x -q > gpurun_out/r02_gpu_tests15.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r02_gpu_tests15.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r02_gpu_tests15.log | head -30; exit 1; }
timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null
