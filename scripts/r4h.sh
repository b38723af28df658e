mkdir -p gpurun_out/r4h
python scripts/bench_sweep_quad.py > gpurun_out/r4h/quad.txt 2>&1
BMV_QUAD_DEFS="-DBMV_QUAD_TAPBUF=2" python -m boostmvsnerfs_amd.build > /dev/null 2>&1
python scripts/bench_sweep_quad.py > gpurun_out/r4h/quad_tapbuf2.txt 2>&1
