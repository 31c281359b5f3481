#!/bin/bash
# round-2 GPU call 1: whole GPU suite with the new tests, then AR loop A/B (head launch with 8 vs 4 waves)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02_gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r02_gpu_tests.log
[ $rc -ne 0 ] && exit $rc
TTK_AR_WV_HEAD=8 timeout -k 10 120 python tests/diag/ar_ab.py 5 > gpurun_out/r02_arab.log 2>&1 && \
TTK_AR_WV_HEAD=4 timeout -k 10 120 python tests/diag/ar_ab.py 5 >> gpurun_out/r02_arab.log 2>&1 && \
TTK_AR_WV_HEAD=8 timeout -k 10 120 python tests/diag/ar_ab.py 5 >> gpurun_out/r02_arab.log 2>&1 && \
TTK_AR_WV_HEAD=4 timeout -k 10 120 python tests/diag/ar_ab.py 5 >> gpurun_out/r02_arab.log 2>&1
cat gpurun_out/r02_arab.log
