mkdir -p gpurun_out/r4z2
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4z2/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4z2/pytest_gpu.txt | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4z2/smoke.txt 2>&1; tail -2 gpurun_out/r4z2/smoke.txt
timeout 900 python bench.py > gpurun_out/r4z2/bench_default.json 2> gpurun_out/r4z2/bench_default.err
