#!/bin/bash
# round-5: MFMA / VALU co-issue probe; deterministic mode after the magnitude pass stopped hammering its slots
R=$(pwd); O=$R/gpurun_out/r5o; mkdir -p $O
$R/scripts/ubench/mfma_valu_coissue > $O/mfma_valu_coissue.txt 2>&1
cat $O/mfma_valu_coissue.txt
python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -q -x 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
for d in 0 1; do
  BMV_DETERMINISTIC=$d python3 bench.py --workload enerf_ft_512x640_3src --steps 16 --warmup 6 --no-cpu-baseline > $O/ft_det$d.json 2> $O/ft_det$d.err
  python3 -c "
import json; d=json.loads(open('$O/ft_det$d.json').read().strip().splitlines()[-1]); print('det $d', d['ms_per_step'])"
done
BMV_DETERMINISTIC=1 python3 bench.py --workload enerf_ours_ft_480x736_6src_k4 --steps 6 --warmup 4 --no-cpu-baseline > $O/c5_det1.json 2> $O/c5_det1.err
python3 -c "
import json; d=json.loads(open('$O/c5_det1.json').read().strip().splitlines()[-1]); print('c5 det 1', d['ms_per_step'])"
