set -e
mkdir -p gpurun_out
rm -f gpurun_out/stress_errors.json
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gemm_roles.py tests/test_gpu_fp8.py -x -q > gpurun_out/r06_fp8_tests.log 2>&1 || { tail -40 gpurun_out/r06_fp8_tests.log; exit 1; }
tail -2 gpurun_out/r06_fp8_tests.log
TTK_TEST_ALL_MODES=1 python -m pytest tests/test_gpu_stress.py -q -k "diffusion or loop" > gpurun_out/r06_stress_all.log 2>&1 || { tail -60 gpurun_out/r06_stress_all.log; }
tail -3 gpurun_out/r06_stress_all.log
python bench.py --dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_fp8_a.json 2> gpurun_out/r06_bench_fp8_a.err || { tail -30 gpurun_out/r06_bench_fp8_a.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_fp8_a.json').read().strip().splitlines()[-1]); print('fp8', d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['roofline']['phases'].items() if isinstance(v,dict) and 'ms' in v})"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err || { tail -30 gpurun_out/r06_bench_b.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_b.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['roofline']['phases'].items() if isinstance(v,dict) and 'ms' in v})"
