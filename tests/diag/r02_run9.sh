#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for o in 0 1 2 3 0 1; do TTK_GEMM_ORDER=$o timeout -k 10 120 python tests/diag/ddim_ab.py 3 2>/dev/null; done > gpurun_out/r02_gemm_order.log; cat gpurun_out/r02_gemm_order.log
