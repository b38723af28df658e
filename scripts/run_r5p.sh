#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5p; mkdir -p $O
$R/scripts/ubench/mfma_valu_coissue > $O/mfma_valu_coissue.txt 2>&1
cat $O/mfma_valu_coissue.txt
