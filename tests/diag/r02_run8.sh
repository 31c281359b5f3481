#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=4 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=16 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_NARROW=2 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_NARROW2=2 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_ATTN_DECODE=2 TTK_ATTN_DECODE=2 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AB_DTYPE=f32 timeout -k 10 120 python tests/diag/ar_ab.py 2 ) 2>/dev/null > gpurun_out/r02_arab8.log
cat gpurun_out/r02_arab8.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_bench_shapes.py tests/test_gpu_fp8.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02_gpu_tests8.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r02_gpu_tests8.log
