#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( TTK_ATTN_W8=0 timeout -k 10 120 python tests/diag/ddim_ab.py 3
  TTK_ATTN_W8=1 timeout -k 10 120 python tests/diag/ddim_ab.py 3
  TTK_ATTN_W8=1 TTK_GN_PASSES=1 timeout -k 10 120 python tests/diag/ddim_ab.py 3
  TTK_ATTN_W8=0 timeout -k 10 120 python tests/diag/ddim_ab.py 3 ) 2>/dev/null > gpurun_out/r02_ddim10.log; cat gpurun_out/r02_ddim10.log
timeout -k 10 600 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_cond.py -m gpu -x -q -k "diff or T1088 or T2176 or ddim or evaluation or cond or odd" > gpurun_out/r02_gpu_tests10.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r02_gpu_tests10.log
