bash scripts/pmc_sweep.sh gpurun_out/r4x > gpurun_out/r4x_pmc.log 2>&1
bash scripts/pmc_sweep_sq.sh gpurun_out/r4x 0,4 >> gpurun_out/r4x_pmc.log 2>&1
