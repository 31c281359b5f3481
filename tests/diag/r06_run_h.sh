set -e
mkdir -p gpurun_out
OUT=gpurun_out/r06_attn_2slot.log
: > $OUT
CASES="bf16:1088 bf16:1000 f16:1088 bf16:800 bf16:1150 bf16:320 bf16:2176 f32:320"
TTK_LIB=$PWD/tortoise_tts_amd/libttk_base.so timeout -k 10 400 python tests/diag/lib_bits.py $CASES 2>/dev/null > gpurun_out/r06_attn_bits_base.txt
timeout -k 10 400 python tests/diag/lib_bits.py $CASES 2>/dev/null > gpurun_out/r06_attn_bits_2slot.txt
if diff gpurun_out/r06_attn_bits_base.txt gpurun_out/r06_attn_bits_2slot.txt >> $OUT; then echo "two-slot form == single-slot form, bit for bit: $CASES" >> $OUT; else echo "DIFFERENT" >> $OUT; fi
echo "== chain T=1088 random, two K/V slots, one barrier per key tile" >> $OUT
DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain.bin >> $OUT 2>&1
grep -v "^gn_apply\|^gemm\|^side\|clock, gemm" $OUT
