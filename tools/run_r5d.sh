#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
O=gpurun_out/r5d; mkdir -p $O
GAP_MS=8 REPS=4 rocprofv3 --kernel-trace --output-format csv -d $O/prove_trace -- python3 tools/prove_loop.py > $O/prove_loop.log 2>&1
python3 tools/prove_gaps.py $(find $O/prove_trace -name "*kernel_trace.csv" | head -1) > $O/prove_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
tail -40 $O/prove_timeline.txt
