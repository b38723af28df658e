mkdir -p gpurun_out/r5h
python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_parity.py tests/test_gpu_boost.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5h/tests.txt
for v in 1 0; do
BMV_FEED_IN_SETUP=$v python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feed in setup $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), d['value_extra']['step_ms'])" >> gpurun_out/r5h/feed_ab.txt
done
python bench.py --workload enerf_256x320_3src_32planes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), {k:round(v['frac'],3) for k,v in d['roofline']['levels'].items()})" >> gpurun_out/r5h/feed_ab.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/hof && rocprofv3 --kernel-trace --output-format csv -d /tmp/hof -- python3 $GRAFT_REPO_ROOT/scripts/head_of_frame.py > /tmp/hof.txt 2>&1; T=$(ls /tmp/hof/*/*kernel_trace.csv | head -1); python3 $GRAFT_REPO_ROOT/scripts/frame_timeline.py $T --frame -40 | head -12 > $GRAFT_REPO_ROOT/gpurun_out/r5h/head_fresh.txt; python3 $GRAFT_REPO_ROOT/scripts/frame_timeline.py $T --frame -3 | head -12 > $GRAFT_REPO_ROOT/gpurun_out/r5h/head_resident.txt)
cat gpurun_out/r5h/tests.txt gpurun_out/r5h/feed_ab.txt gpurun_out/r5h/head_fresh.txt gpurun_out/r5h/head_resident.txt
