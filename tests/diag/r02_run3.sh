#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q --durations=8 > gpurun_out/r02_gpu_tests3.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r02_gpu_tests3.log
bash tests/diag/ar_ablate.sh run > gpurun_out/r02_ar_ablate.log 2>&1; cat gpurun_out/r02_ar_ablate.log
for p in 2 4 8 2 8; do TTK_GN_PASSES=$p timeout -k 10 120 python tests/diag/ddim_ab.py 3 2>/dev/null; done > gpurun_out/r02_gn_passes.log; cat gpurun_out/r02_gn_passes.log
timeout -k 10 200 python tests/diag/ddim_graph.py 3 > gpurun_out/r02_ddim_graph.log 2>&1; tail -5 gpurun_out/r02_ddim_graph.log
