mkdir -p gpurun_out/r4z3
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4z3/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4z3/pytest_gpu.txt | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4z3/smoke.txt 2>&1; tail -1 gpurun_out/r4z3/smoke.txt
bash scripts/collect_profiles.sh gpurun_out/r4z3 c2 c1 c3 c4 c5 ft > gpurun_out/r4z3/collect.log 2>&1
timeout 900 python bench.py > gpurun_out/r4z3/bench_default.json 2> gpurun_out/r4z3/bench_default.err
timeout 600 python scripts/probe_autograph_cost.py > gpurun_out/r4z3/probe.txt 2>&1
