mkdir -p gpurun_out/r4s
python tests/tools/grad_fp64_arbitration.py --gpu > gpurun_out/r4s/arb.txt 2>&1
