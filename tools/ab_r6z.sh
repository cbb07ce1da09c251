#!/bin/bash
# round 6, call Z: idle GPU inside native proofs -- host columns (typlonk_prove_host) at 2^20, device columns at 2^22
export TMPDIR=/tmp
O=gpurun_out/r6z; rm -rf $O; mkdir -p $O
for v in "1 20" "0 22"; do set -- $v
  HOST=$1 LOG_N=$2 GAP_MS=8 REPS=3 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1_$2 -- python3 tools/prove_native_loop.py > $O/loop_$1_$2.log 2>&1
  python3 tools/trace_timeline.py $(find $O/trace_$1_$2 -name "*kernel_trace.csv" | head -1) > $O/timeline_$1_$2.txt 2>&1
  echo "== HOST=$1 LOG_N=$2"; grep "idle" $O/timeline_$1_$2.txt | cut -c1-120; tail -n 1 $O/timeline_$1_$2.txt; tail -n 1 $O/loop_$1_$2.log
done
find $O -name "*kernel_trace.csv" -size +1M -delete
