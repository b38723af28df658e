mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4w/ta.txt 2>&1; tail -3 gpurun_out/r4w/ta.txt | cut -c1-180
run() { echo "== $*"; env "$@" timeout 300 python scripts/probe_autograph_cost.py 2>&1 | grep -a "resident_inputs=True  alias_outputs=True\|resident_inputs=False alias_outputs=False" | cut -c1-75; }
{
run BMV_SETUP_SIDE=0
run BMV_SETUP_SIDE=1
run BMV_SETUP_SIDE=0
run BMV_SETUP_SIDE=1
run BMV_SETUP_SIDE=0
run BMV_SETUP_SIDE=1
} > gpurun_out/r4w/knobs.txt 2>&1
