#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02_gpu_tests21.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -8 gpurun_out/r02_gpu_tests21.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|^E " gpurun_out/r02_gpu_tests21.log | head -40; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
