#!/bin/bash
touch boostmvsnerfs_amd/csrc/render.hip
BMV_RENDER_DEFS="-DBMV_RENDER_PC_COUNT" python -m boostmvsnerfs_amd.build 2>&1 | grep -i " error"
timeout 300 python scripts/pc_spins.py 2>&1 | tail -4
