mkdir -p gpurun_out/r5f
python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r5f/pytest_gpu.txt
python bench.py > gpurun_out/r5f/bench_config2_c4.json 2> gpurun_out/r5f/bench_config2_c4.err
BMV_CONV_C4=0 python bench.py --no-cpu-baseline > gpurun_out/r5f/bench_config2_noc4.json 2> gpurun_out/r5f/bench_config2_noc4.err
python tests/tools/config5_grad_probe.py 16 --fp64 > gpurun_out/r5f/c5_probe_fp64.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/eg -o eg -- $GRAFT_REPO_ROOT/scripts/ubench/empty_grid > $GRAFT_REPO_ROOT/gpurun_out/r5f/empty_grid.txt 2>&1; find /tmp/eg -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r5f/empty_grid_kernel_stats.csv \;)
cat gpurun_out/r5f/pytest_gpu.txt | tail -8
python - <<'P'
import json
for f in ('gpurun_out/r5f/bench_config2_c4.json','gpurun_out/r5f/bench_config2_noc4.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],2), round(d['ms_per_step'],4), 'resident', d['value_extra'].get('resident_batch',{}).get('value'), 'parity', (d.get('parity_max_rel') or {}).get('max'), 'roofline', {k:round(v['frac'],3) for k,v in d['roofline']['levels'].items()})
    except Exception as e: print(f,'ERR',e)
P
grep -v "VGG\|amdgpu\|Warn\|warn\|print(" gpurun_out/r5f/c5_probe_fp64.txt | tail -60
cat gpurun_out/r5f/empty_grid.txt | tail -5; cat gpurun_out/r5f/empty_grid_kernel_stats.csv 2>/dev/null | head -12
