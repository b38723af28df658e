mkdir -p gpurun_out/r5l
for wl in enerf_ours_480x736_6src_k4 enerf_256x320_3src_32planes enerf_512x640_2src_64planes enerf_512x640_4src_64planes; do
python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(d['value'],2), round(d['ms_per_step'],4), 'resident', (d['value_extra'].get('resident_batch') or {}).get('value'))" >> gpurun_out/r5l/lines.txt
done
python bench.py --workload enerf_ours_ft_480x736_6src_k4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5', round(d['value'],3), round(d['ms_per_step'],3))" >> gpurun_out/r5l/lines.txt
python bench.py --workload enerf_ft_512x640_3src --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enerf_ft', round(d['value'],3), round(d['ms_per_step'],3))" >> gpurun_out/r5l/lines.txt
cat gpurun_out/r5l/lines.txt
