#!/bin/bash
show() { python -c "
import json,sys
d=json.loads(open('$1').read())
k=[v for n,v in d['kernels'].items() if n.startswith('render_rays')]
print('$2 value %.1f ms %.4f' % (d['value'], d['ms_per_step']), '| render us', [round(x['avg_us'],1) for x in k])
"; }
run() {
  touch boostmvsnerfs_amd/csrc/render.hip
  BMV_RENDER_DEFS="$1" python -m boostmvsnerfs_amd.build 2>&1 | grep -i " error"
  timeout 600 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 > /tmp/b.json; show /tmp/b.json "$1"
}
run "-DBMV_RENDER_PC_GATHER=4 -DBMV_RENDER_PC_FAKE_GATHER"
run "-DBMV_RENDER_PC_GATHER=8 -DBMV_RENDER_PC_FAKE_GATHER"
