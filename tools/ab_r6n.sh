#!/bin/bash
# round 6, call N: planes kernel with four-lane steps across its wavefronts -- parity + timeline
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_host_mirror.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline.txt 2>&1; cat $O/msm_timeline.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
for rep in 1 2 3; do python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done
WORLD=8 TABLES=auto REPS=40 python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" | cut -c1-330
