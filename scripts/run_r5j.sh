mkdir -p gpurun_out/r5j
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r5j/pytest_gpu.txt
for v in 1 0; do
BMV_DEPTH_MAPS_TABLE=$v python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth maps through table $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), 'median', round(d['value_extra']['step_ms']['median'],4))" >> gpurun_out/r5j/ab.txt
done
cat gpurun_out/r5j/pytest_gpu.txt gpurun_out/r5j/ab.txt
