#!/bin/bash
# kernel timeline of one proof with its accumulation-free stretches (rocprofv3 --kernel-trace + tools/prove_gaps.py); run on the GPU box
export TMPDIR=/tmp
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
GAP_MS=8 REPS=4 rocprofv3 --kernel-trace --output-format csv -d $O/prove_trace -- python3 tools/prove_loop.py > $O/prove_loop.log 2>&1
python3 tools/prove_gaps.py $(find $O/prove_trace -name "*kernel_trace.csv" | head -1) > $O/prove_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
sed -n 1,75p $O/prove_timeline.txt | cut -c1-90; grep -A12 "accumulation-free" $O/prove_timeline.txt | cut -c1-150
