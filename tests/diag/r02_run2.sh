#!/bin/bash
# round-2 GPU call 2: two-queue flag chain microbenchmark, then the tests added since call 1
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 150 tests/diag/twoq.bin 600 4 > gpurun_out/r02_twoq.log 2>&1; echo "rc $?" >> gpurun_out/r02_twoq.log
timeout -k 10 150 tests/diag/twoq.bin 600 1 >> gpurun_out/r02_twoq.log 2>&1; echo "rc $?" >> gpurun_out/r02_twoq.log
timeout -k 10 150 tests/diag/twoq.bin 600 8 >> gpurun_out/r02_twoq.log 2>&1; echo "rc $?" >> gpurun_out/r02_twoq.log
cat gpurun_out/r02_twoq.log
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q --durations=8 > gpurun_out/r02_gpu_tests2.log 2>&1
rc=$?
tail -15 gpurun_out/r02_gpu_tests2.log
exit $rc
