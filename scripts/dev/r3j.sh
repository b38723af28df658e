#!/bin/bash
mkdir -p gpurun_out/r3j
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)[^,]*" | head -5
rocprofv3 -L 2>/dev/null | grep -E "TA_|TCP_|TD_|TCC_(REQ|READ|WRITE|HIT|MISS|EA)|LDS|GRBM_GUI" | cut -c1-160 | head -120 > gpurun_out/r3j/counters.txt
wc -l gpurun_out/r3j/counters.txt; head -100 gpurun_out/r3j/counters.txt
