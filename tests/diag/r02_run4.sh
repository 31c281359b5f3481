#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02_gpu_tests4.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r02_gpu_tests4.log
( TTK_AR_LNFOLD=0 TTK_AR_SHARE_PREFIX=0 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_LNFOLD=1 TTK_AR_SHARE_PREFIX=0 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_LNFOLD=0 TTK_AR_SHARE_PREFIX=1 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_LNFOLD=1 TTK_AR_SHARE_PREFIX=1 timeout -k 10 120 python tests/diag/ar_ab.py 3 ) 2>/dev/null > gpurun_out/r02_arab4.log
cat gpurun_out/r02_arab4.log
