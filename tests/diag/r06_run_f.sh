set -e
mkdir -p gpurun_out
OUT=gpurun_out/r06_attn_ring.log
: > $OUT
CASES="bf16:1088 bf16:1000 f16:1088 bf16:800 bf16:1150 bf16:784"
TTK_ATTN_RING=0 timeout -k 10 300 python tests/diag/lib_bits.py $CASES > gpurun_out/r06_attn_bits_barrier.txt 2>&1
TTK_ATTN_RING=1 timeout -k 10 300 python tests/diag/lib_bits.py $CASES > gpurun_out/r06_attn_bits_ring.txt 2>&1
if diff gpurun_out/r06_attn_bits_barrier.txt gpurun_out/r06_attn_bits_ring.txt >> $OUT; then echo "ring == barrier form, bit for bit: $CASES" >> $OUT; else echo "DIFFERENT" >> $OUT; fi
cat gpurun_out/r06_attn_bits_ring.txt >> $OUT
echo "== chain T=1088 random, barrier form" >> $OUT
TTK_ATTN_RING=0 DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain.bin >> $OUT 2>&1
echo "== chain T=1088 random, ring" >> $OUT
TTK_ATTN_RING=1 DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain.bin >> $OUT 2>&1
grep -v "^gn_apply\|^gemm\|^side" $OUT
