#!/bin/bash
# round 6, call F: four-lane butterfly steps in the reduction, queued MSMs above 2^20 terms in chunks -- FULL gpu suite, A/B
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2 3; do
  echo "== new rep $rep"; python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch new rep $rep"; REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
  echo "== r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
done > $O/msm_ab.txt 2>&1
cat $O/msm_ab.txt
REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline.txt 2>&1; cat $O/msm_timeline.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
for rep in 1 2; do echo "== new"; python3 tools/prove_rounds.py 2>/dev/null | tail -2; echo "== r5base"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/prove_rounds.py 2>/dev/null | tail -2; done > $O/prove_ab.txt 2>&1
cat $O/prove_ab.txt
for rep in 1 2; do echo "== 2^22 new"; LOG_N=22 python3 tools/prove_rounds.py 2>/dev/null | tail -2; echo "== 2^22 r5base"; LOG_N=22 TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/prove_rounds.py 2>/dev/null | tail -2; done > $O/prove_ab_22.txt 2>&1
cat $O/prove_ab_22.txt
echo "== batch 2^22 new"; LOG_M=22 REPS=4 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1; echo "== batch 2^22 r5base"; LOG_M=22 REPS=4 TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
WORLD=8 TABLES=auto REPS=40 python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" > $O/shard8.txt; WORLD=8 TABLES=auto REPS=40 TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" >> $O/shard8.txt; cat $O/shard8.txt
