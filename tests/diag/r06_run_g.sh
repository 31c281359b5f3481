set -e
OUT=gpurun_out/r06_attn_ring_skew.log
: > $OUT
for K in 10 20 40; do
echo "== ring, second wave of each SIMD starts $K x 64 cycles late" >> $OUT
TTK_ATTN_RING=1 DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain_skew$K.bin >> $OUT 2>&1
done
grep "^==\|^attention\|clock, attention\|ddim chain" $OUT
