mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k config5 -s 2>&1 | grep -E "config 5|passed|failed|Error|assert" | head -20 > gpurun_out/r5g/c5_test.txt
python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r5g/conv_tests.txt
for rows in 4 2 1 0; do
  BMV_CONV_SPLITK_ROWS=$rows python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splitk rows $rows', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2))" >> gpurun_out/r5g/splitk_rows.txt
done
bash scripts/run_timeline.sh gpurun_out/r5g/timeline > /dev/null 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eg -o eg -- $GRAFT_REPO_ROOT/scripts/ubench/empty_grid > /tmp/eg.txt 2>&1; find /tmp/eg -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r5g/empty_grid_kernel_stats.csv \;)
cat gpurun_out/r5g/c5_test.txt gpurun_out/r5g/conv_tests.txt gpurun_out/r5g/splitk_rows.txt
cat gpurun_out/r5g/empty_grid_kernel_stats.csv
cat gpurun_out/r5g/timeline/frame_overlap2.txt
