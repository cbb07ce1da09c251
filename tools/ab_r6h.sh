#!/bin/bash
# round 6, call H: LDS-staged place kernel, prover NTT modes 0/2/3 at 2^20 and 2^22, the dist tests again
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_prove.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline.txt 2>&1; cat $O/msm_timeline.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
for rep in 1 2 3; do for v in 0 2 3; do echo "== TYPLONK_PROVER_NTT_BATCH=$v rep $rep"; TYPLONK_PROVER_NTT_BATCH=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; done > $O/prove_ab.txt 2>&1; cat $O/prove_ab.txt
for rep in 1 2; do for v in 0 2 3; do echo "== 2^22 TYPLONK_PROVER_NTT_BATCH=$v rep $rep"; LOG_N=22 TYPLONK_PROVER_NTT_BATCH=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; done > $O/prove_ab_22.txt 2>&1; cat $O/prove_ab_22.txt
for rep in 1 2 3; do
  echo "== new rep $rep"; python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch new rep $rep"; REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
  echo "== r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
done > $O/msm_ab.txt 2>&1
cat $O/msm_ab.txt
