#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 100 python tests/diag/rng_probe.py 2>&1 | grep "equal"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02_gpu_tests13.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -6 gpurun_out/r02_gpu_tests13.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r02_gpu_tests13.log | head -30; exit 1; }
timeout -k 10 120 python tests/diag/ar_ab.py 3 2>/dev/null &&
TTK_AR_OWN_RNG=0 timeout -k 10 120 python tests/diag/ar_ab.py 3 2>/dev/null &&
TTK_DIFF_IGROUP=1 timeout -k 10 120 python tests/diag/ddim_ab.py 3 2>/dev/null &&
TTK_DIFF_IGROUP=4 timeout -k 10 120 python tests/diag/ddim_ab.py 3 2>/dev/null &&
TTK_DIFF_IGROUP=8 timeout -k 10 120 python tests/diag/ddim_ab.py 3 2>/dev/null
