#!/bin/bash
# round 6, call Y: kernel timeline of a NATIVE proof (typlonk_prove) -- where is the GPU idle?
export TMPDIR=/tmp
O=gpurun_out/r6y; rm -rf $O; mkdir -p $O
GAP_MS=8 REPS=4 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/prove_native_loop.py > $O/loop.log 2>&1
python3 tools/trace_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
grep "idle" $O/timeline.txt | cut -c1-120; tail -n 1 $O/timeline.txt; tail -n 1 $O/loop.log
REPS=10 python3 tools/prove_native_loop.py 2>/dev/null | tail -n 1
