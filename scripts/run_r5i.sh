mkdir -p gpurun_out/r5i
python -m pytest tests/test_gpu_framegraph.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r5i/tests.txt
for v in 1 0; do
BMV_FEED_IN_SETUP=$v python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feed in setup $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), 'median', round(d['value_extra']['step_ms']['median'],4))" >> gpurun_out/r5i/feed_ab.txt
done
for v in 1 0; do
BMV_CONV_SPLITK_PF=$v python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splitk prefetch $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), 'median', round(d['value_extra']['step_ms']['median'],4))" >> gpurun_out/r5i/feed_ab.txt
done
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/hof && rocprofv3 --kernel-trace --output-format csv -d /tmp/hof -- python3 $GRAFT_REPO_ROOT/scripts/head_of_frame.py > /tmp/hof.txt 2>&1; T=$(ls /tmp/hof/*/*kernel_trace.csv | head -1); python3 $GRAFT_REPO_ROOT/scripts/frame_timeline.py $T --frame -40 | head -6 > $GRAFT_REPO_ROOT/gpurun_out/r5i/head_fresh.txt)
cat gpurun_out/r5i/tests.txt gpurun_out/r5i/feed_ab.txt gpurun_out/r5i/head_fresh.txt
