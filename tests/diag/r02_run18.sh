#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_gemm.py tests/test_gpu_parity.py "tests/test_gpu_bench_shapes.py::test_config1_teacher_forced_decode_full_size" "tests/test_gpu_bench_shapes.py::test_config1_evaluation_and_ddim_slice_at_T1088" -m gpu -x -q > gpurun_out/r02_gpu_tests18.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r02_gpu_tests18.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|^E " gpurun_out/r02_gpu_tests18.log | head -40; exit 1; }
timeout -k 10 300 python bench.py --dtype f16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_f16.json 2> gpurun_out/r02_bench_f16.log
echo "bench rc $?"; tail -c 600 gpurun_out/r02_bench_f16.json
