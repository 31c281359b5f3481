#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( timeout -k 5 60 tests/diag/skinny_stamps.bin 4 0; timeout -k 5 60 tests/diag/skinny_stamps.bin 4 1 ) > gpurun_out/r02_stamps2.log 2>&1; cat gpurun_out/r02_stamps2.log
( TTK_AR_WV_PROJ2=8 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=4 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=8 TTK_AR_WV_PROJ=4 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=8 TTK_AR_WV_PROJ=16 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=8 TTK_AR_WV_LN=4 timeout -k 10 120 python tests/diag/ar_ab.py 3
  TTK_AR_WV_PROJ2=8 TTK_AR_WV_LN=16 timeout -k 10 120 python tests/diag/ar_ab.py 3 ) 2>/dev/null > gpurun_out/r02_arab7.log
cat gpurun_out/r02_arab7.log
