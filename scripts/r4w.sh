bash scripts/run_timeline.sh gpurun_out/r4w_tl > gpurun_out/r4w_tl.log 2>&1
