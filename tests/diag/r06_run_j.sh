set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_2.log 2>&1 || { tail -80 gpurun_out/r06_gputest_2.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r06_gputest_2.log
bash profiles/collect.sh r06 2>&1 | tail -5
python bench.py --dtype fp8 --no-cpu-baseline > gpurun_out/r06_bench_fp8.json 2> gpurun_out/r06_bench_fp8.err || { tail -30 gpurun_out/r06_bench_fp8.err; exit 1; }
python -c "
import json
for f in ('gpurun_out/bench_r06.json','gpurun_out/r06_bench_fp8.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['roofline']['phases'].items() if isinstance(v,dict) and 'ms' in v})"
