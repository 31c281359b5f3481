set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_1.log 2>&1 || { tail -80 gpurun_out/r06_gputest_1.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r06_gputest_1.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c.json 2> gpurun_out/r06_bench_c.err || { tail -30 gpurun_out/r06_bench_c.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_c.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['roofline']['phases'].items() if isinstance(v,dict) and 'ms' in v})"
