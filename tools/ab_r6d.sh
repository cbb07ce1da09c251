#!/bin/bash
# round 6, call D: replicated schedule counters, fold_seq on 4-wavefront workgroups, level-1 workgroup size -- parity, A/B
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_prove.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2 3; do
  for t in 256 512; do
  echo "== new L1=$t rep $rep"; TYPLONK_MSM_L1_THREADS=$t python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch new L1=$t rep $rep"; TYPLONK_MSM_L1_THREADS=$t REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
  done
  echo "== r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
done > $O/msm_ab.txt 2>&1
cat $O/msm_ab.txt
for t in 256 512; do
TYPLONK_MSM_L1_THREADS=$t REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace$t -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace$t -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline_$t.txt 2>&1; cat $O/msm_timeline_$t.txt
done
find $O -name "*kernel_trace.csv" -size +4M -delete
for rep in 1 2; do for t in 256 512; do echo "== new L1=$t"; TYPLONK_MSM_L1_THREADS=$t python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; echo "== r5base"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/prove_rounds.py 2>/dev/null | tail -2; done > $O/prove_ab.txt 2>&1
cat $O/prove_ab.txt
WORLD=8 TABLES=auto REPS=40 python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" > $O/shard8.txt; WORLD=8 TABLES=auto REPS=40 TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" >> $O/shard8.txt; cat $O/shard8.txt
