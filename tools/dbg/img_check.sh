python -m pytest tests/test_snet_dt_gpu.py tests/test_mlp_gpu.py tests/test_calculators_gpu.py tests/test_training_options_gpu.py tests/test_bench_gpu.py -m gpu -x -q > gpurun_out/img_check_tests.log 2>&1; tail -3 gpurun_out/img_check_tests.log
python bench.py --config c2 --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2), round(d['roofline']['avg_ms']*1e3,2))"
python bench.py --config ref_small --steps 300 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']: print(r['network'], r['batch'], round(r['value']/1e6,2), round(r['us_per_step'],1))"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2)); print({k:(round(v['value']/1e6,2) if isinstance(v,dict) and 'value' in v else None) for k,v in d.items() if k in ('shuffled','c2','ref_small')})"
