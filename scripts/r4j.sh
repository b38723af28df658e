mkdir -p gpurun_out/r4j
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r4j/pytest.txt 2>&1
tail -15 gpurun_out/r4j/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err
