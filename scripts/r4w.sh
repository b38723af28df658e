mkdir -p gpurun_out/r4w
for sp in 1 2 4; do
BMV_FPN_TOPDOWN_SPLIT=$sp timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "fpn" 2>&1 | tail -1 | cut -c1-100
BMV_FPN_TOPDOWN_SPLIT=$sp timeout 600 python scripts/probe_autograph_cost.py 2>&1 | grep -a "resident_inputs=True  alias_outputs=True\|resident_inputs=False alias_outputs=False" | cut -c1-80
done > gpurun_out/r4w/split.txt 2>&1
