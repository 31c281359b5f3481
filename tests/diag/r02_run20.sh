#!/bin/bash
set -o pipefail
for i in 1 2; do
timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null &&
TTK_LIB=$PWD/tortoise_tts_amd/libttk_attn.so timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null &&
TTK_LIB=$PWD/tortoise_tts_amd/libttk_attn_p.so timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null &&
TTK_LIB=$PWD/tortoise_tts_amd/libttk_attn_v.so timeout -k 10 120 python tests/diag/ddim_ab.py 5 2>/dev/null || exit 1
done
