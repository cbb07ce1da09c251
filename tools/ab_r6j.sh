#!/bin/bash
# round 6, call J: hardware queues (GPU_MAX_HW_QUEUES) x MSMs in flight
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
for rep in 1 2; do for q in 4 8; do for v in 3 4 5 6; do echo "== HWQ=$q INFLIGHT=$v rep $rep"; GPU_MAX_HW_QUEUES=$q TYPLONK_MSM_INFLIGHT=$v REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1; done; done; done > $O/batch_ab.txt 2>&1; cat $O/batch_ab.txt
for rep in 1 2; do for q in 4 8; do for v in 3 4 5; do echo "== prove HWQ=$q INFLIGHT=$v rep $rep"; GPU_MAX_HW_QUEUES=$q TYPLONK_MSM_INFLIGHT=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; done; done > $O/prove_ab.txt 2>&1; cat $O/prove_ab.txt
GPU_MAX_HW_QUEUES=8 TYPLONK_MSM_INFLIGHT=5 REPS=3 BATCH=9 rocprofv3 --kernel-trace --output-format csv -d $O/batch_trace -- python3 tools/msm_batch_loop.py > $O/batch_loop.log 2>&1
python3 tools/trace_timeline.py $(find $O/batch_trace -name "*kernel_trace.csv" | head -1) > $O/msm_batch_timeline_q8_5.txt 2>&1; grep -n "accum\|span" $O/msm_batch_timeline_q8_5.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
