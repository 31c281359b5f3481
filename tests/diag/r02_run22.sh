#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fp16.py "tests/test_gpu_bench_shapes.py::test_config1_evaluation_and_ddim_slice_at_T1088" "tests/test_gpu_bench_shapes.py::test_config3_evaluation_at_T2176" -m gpu -x -q > gpurun_out/r02_gpu_tests22.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r02_gpu_tests22.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|^E " gpurun_out/r02_gpu_tests22.log | head -40; exit 1; }
for i in 1 2; do
timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null &&
TTK_ATTN_W3=0 timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null || exit 1
done
