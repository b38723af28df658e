mkdir -p gpurun_out/r4k
for mb in 0 16 32 64 128; do echo "== evict $mb MB"; python scripts/bench_sweep_quad.py --evict-mb $mb --variants 0 2>&1 | grep "level 1"; python scripts/bench_sweep_quad.py --evict-mb $mb --variants 12 2>&1 | grep "level 0"; done > gpurun_out/r4k/evict.txt 2>&1
