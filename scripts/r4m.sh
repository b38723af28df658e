mkdir -p gpurun_out/r4m
python scripts/bench_sweep_quad.py --variants 12 2>&1 | grep "level 0" > gpurun_out/r4m/l0.txt
python scripts/bench_sweep_quad.py --variants 12 --flags 8 2>&1 | grep "level 0" >> gpurun_out/r4m/l0.txt
for f in 1 2 4 7; do python scripts/bench_sweep_quad.py --variants 12 --flags $f 2>&1 | grep "level 0  quad" >> gpurun_out/r4m/l0.txt; done
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4m/bench.json 2> gpurun_out/r4m/bench.err
