mkdir -p gpurun_out/r4p
python scripts/probe_mask_share.py > gpurun_out/r4p/mask_share.txt 2>&1
