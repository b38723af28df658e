mkdir -p gpurun_out/r5k
python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r5k/conv_tests.txt
python scripts/bench_conv_c4.py > gpurun_out/r5k/c4_bench.txt 2>&1
python - > gpurun_out/r5k/convT_bench.txt 2>&1 <<'P'
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'scripts')
from boostmvsnerfs_amd import convnet
from bench_conv_c4 import timed
for name, D, H, W in (("L1 conv11 16->8", 4, 128, 160), ("L0 conv11 16->8", 32, 32, 40)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, D, H, W, generator=g).cuda()
    w = (torch.randn(16, 8, 3, 3, 3, generator=g) / 8).cuda()
    b = torch.randn(8, generator=g).cuda()
    skip = torch.randn(1, 8, 2 * D, 2 * H, 2 * W, generator=g).cuda()
    wp16, bp16 = convnet.pack_convT(w, b)
    wp4, bp4 = convnet.pack_convT_c4(w, b)
    t16 = timed(lambda: convnet.convT3d_fwd(x, wp16, bp16, 8, skip=skip))
    line = f"{name}: engine {t16:6.1f} us"
    for v in (0, 1, 2):
        line += f"  c4 v{v} {timed(lambda: convnet.convT_c4_fwd(x, wp4, bp4, 8, skip=skip, variant=v)):6.1f} us"
    print(line)
P
for v in 1 0; do
BMV_CONV_C4=$v python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv_c4 $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), 'median', round(d['value_extra']['step_ms']['median'],4))" >> gpurun_out/r5k/ab.txt
done
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_boost.py -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/r5k/conv_tests.txt
cat gpurun_out/r5k/conv_tests.txt gpurun_out/r5k/c4_bench.txt gpurun_out/r5k/convT_bench.txt gpurun_out/r5k/ab.txt
