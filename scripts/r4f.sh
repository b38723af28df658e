mkdir -p gpurun_out/r4f
python scripts/bench_sweep_quad.py > gpurun_out/r4f/quad.txt 2>&1
for f in 1 2 4 3 7; do echo "== flags $f"; python scripts/bench_sweep_quad.py --variants 0,3 --flags $f 2>&1 | grep "quad variant" | grep "level 1"; python scripts/bench_sweep_quad.py --variants 1 --flags $f 2>&1 | grep "quad variant" | grep "level 0";  done > gpurun_out/r4f/ablate.txt 2>&1
