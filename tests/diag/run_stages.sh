set -e
cd tests/diag
OUT=../../gpurun_out/r06_stages.log
: > $OUT
echo "== role_check s5 T=1024" >> $OUT; timeout -k 10 120 ./role_check_s5.bin 1024 2 >> $OUT 2>&1
echo "== role_check s5 T=1088" >> $OUT; timeout -k 10 120 ./role_check_s5.bin 1088 2 >> $OUT 2>&1
for S in 3 4 5 6; do
  echo "== stages $S, DC_T=1024, DC_RANDOM=1" >> $OUT
  DC_T=1024 DC_RANDOM=1 timeout -k 10 120 ./ddim_chain_s$S.bin >> $OUT 2>&1
done
for S in 3 5; do
  echo "== stages $S, DC_T=1088, DC_RANDOM=1" >> $OUT
  DC_T=1088 DC_RANDOM=1 timeout -k 10 120 ./ddim_chain_s$S.bin >> $OUT 2>&1
done
