set -e
mkdir -p gpurun_out
OUT=gpurun_out/r06_conv_image_roles.log
: > $OUT
for T in 1088 1024 1216 576 64; do echo "== role_check T=$T nb=2" >> $OUT; timeout -k 10 120 tests/diag/role_check.bin $T 2 >> $OUT 2>&1; done
echo "== role_check T=1088 f8" >> $OUT; timeout -k 10 120 tests/diag/role_check.bin 1088 2 f8 >> $OUT 2>&1
echo "== role_check T=704 nb=3" >> $OUT; timeout -k 10 120 tests/diag/role_check.bin 704 3 >> $OUT 2>&1
grep "role 2\|==" $OUT
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_1.log 2>&1 || { tail -60 gpurun_out/r06_gputest_1.log; exit 1; }
tail -3 gpurun_out/r06_gputest_1.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c.json 2> gpurun_out/r06_bench_c.err || { tail -30 gpurun_out/r06_bench_c.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_c.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['roofline']['phases'].items() if isinstance(v,dict) and 'ms' in v})"
