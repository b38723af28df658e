mkdir -p gpurun_out/r4z
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r4z/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4z/pytest_gpu.txt | cut -c1-200
bash scripts/collect_profiles.sh gpurun_out/r4z c2 c1 c3 c4 c5 ft > gpurun_out/r4z/collect.log 2>&1
timeout 900 python bench.py > gpurun_out/r4z/bench_default.json 2> gpurun_out/r4z/bench_default.err
