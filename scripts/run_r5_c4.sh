#!/bin/bash
PFX=r5 bash scripts/collect_profiles.sh gpurun_out/r5_final/profiles c4 2>&1 | tail -3
bash scripts/run_r5_full_tests2.sh mvs2 2>&1 | tail -14
