#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_boost.py tests/test_gpu_configs34.py tests/test_gpu_framegraph.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -8
show() { python -c "
import json,sys
d=json.loads(open('$1').read())
k=[v for n,v in d['kernels'].items() if n.startswith('render_rays')]
print('$2 value %.1f ms %.4f' % (d['value'], d['ms_per_step']), '| render us', [round(x['avg_us'],1) for x in k], '| mfma frac', round(d['roofline_mfma']['frac'],3))
"; }
for pc in 1 0 1; do
BMV_RENDER_PC=$pc timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3u/b_pc$pc.json; show gpurun_out/r3u/b_pc$pc.json pc=$pc
done
for pc in 1 0; do
BMV_RENDER_PC=$pc timeout 600 python bench.py --workload enerf_ours_480x736_6src_k4 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3u/c3_pc$pc.json; show gpurun_out/r3u/c3_pc$pc.json cfg3_pc=$pc
done
