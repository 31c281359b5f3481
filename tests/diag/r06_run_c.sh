set -e
mkdir -p gpurun_out
OUT=gpurun_out/r06_conv_image.log
: > $OUT
for T in 1088 1024 1216 576; do echo "== role_check T=$T nb=2" >> $OUT; timeout -k 10 120 tests/diag/role_check.bin $T 2 >> $OUT 2>&1; done
echo "== role_check T=1088 f8" >> $OUT; timeout -k 10 120 tests/diag/role_check.bin 1088 2 f8 >> $OUT 2>&1
echo "== chain T=1088 random" >> $OUT
DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain.bin >> $OUT 2>&1
echo "== chain T=1024 random" >> $OUT
DC_T=1024 DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain.bin >> $OUT 2>&1
cat $OUT | grep -v "^gn_apply\|^attention\|^side"
