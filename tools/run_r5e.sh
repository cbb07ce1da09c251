#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
TABLES=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python3 tools/msm_loop.py > $O/plain.log 2>&1
cut -c1-100 $(find $O/plain -name "*kernel_stats.csv" | head -1) | head -24
