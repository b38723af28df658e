mkdir -p gpurun_out/r4y
timeout 900 python scripts/profile_train_ops.py --workload enerf_ours_ft_480x736_6src_k4 --rows 30 > gpurun_out/r4y/ops_c5.txt 2>&1
