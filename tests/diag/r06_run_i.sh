set -e
mkdir -p gpurun_out
OUT=gpurun_out/r06_qkv_pipe.log
: > $OUT
for T in 1088 1024 1216 2176; do echo "== role_check T=$T nb=2" >> $OUT; timeout -k 10 120 tests/diag/role_check.bin $T 2 >> $OUT 2>&1; done
echo "== chain T=1088 random" >> $OUT
DC_RANDOM=1 timeout -k 10 120 tests/diag/ddim_chain.bin >> $OUT 2>&1
grep -v "^gn_apply\|^attention\|^side" $OUT
