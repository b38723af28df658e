mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_fullsize.py tests/test_gpu_backward.py tests/test_gpu_mlp_backward.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5d/pytest_new.txt
for wl in enerf_512x640_2src_64planes enerf_512x640_4src_64planes; do
  python bench.py --workload $wl --steps 200 --no-cpu-baseline > gpurun_out/r5d/bench_$wl.json 2> gpurun_out/r5d/bench_$wl.err
done
python bench.py --workload enerf_ours_480x736_6src_k4 --cpu-baseline > gpurun_out/r5d/bench_config3.json 2> gpurun_out/r5d/bench_config3.err
python bench.py --workload mvsnerf_ours_224x352_128planes_k4 --cpu-baseline --steps 10 > gpurun_out/r5d/bench_config4.json 2> gpurun_out/r5d/bench_config4.err
python bench.py --workload enerf_ours_ft_480x736_6src_k4 --cpu-baseline > gpurun_out/r5d/bench_config5.json 2> gpurun_out/r5d/bench_config5.err
python bench.py > gpurun_out/r5d/bench_config2.json 2> gpurun_out/r5d/bench_config2.err
cat gpurun_out/r5d/pytest_new.txt
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5d/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],2), d['unit'], round(d['ms_per_step'],3),'ms', 'parity', (d.get('parity_max_rel') or {}).get('max'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json','.err')).read()[-1500:])
P
