set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gputest_last.log 2>&1 || { tail -80 gpurun_out/gputest_last.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/gputest_last.log
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_last_$i.json 2> gpurun_out/bench_last_$i.err || { tail -30 gpurun_out/bench_last_$i.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/bench_last_$i.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['roofline']['phases'].items() if isinstance(v,dict) and 'ms' in v})"
done
