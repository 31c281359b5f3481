#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( TTK_AR_HEAD_SPLIT=0 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_HEAD_SPLIT=1 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_HEAD_SPLIT=0 TTK_AR_NARROW2=0 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_HEAD_SPLIT=0 TTK_AR_NARROW=0 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_HEAD_SPLIT=0 TTK_AR_WV_PROJ2=8 timeout -k 10 120 python tests/diag/ar_ab.py 3 ) 2>/dev/null > gpurun_out/r02_arab6.log
cat gpurun_out/r02_arab6.log
timeout -k 10 300 tests/diag/gemm_bench.bin 1 > gpurun_out/r02_gemm_bench.log 2>&1
grep -E "1x1 conv|proj\+res|conv3|qkv|integ" gpurun_out/r02_gemm_bench.log
