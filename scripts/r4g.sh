bash scripts/pmc_sweep_sq.sh gpurun_out/r4g 4,500,503,505,601 > gpurun_out/r4g_pmc.log 2>&1
