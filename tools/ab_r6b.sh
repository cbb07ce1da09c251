#!/bin/bash
# round 6, call B: the rebuilt bucket sort -- parity first, then same-box A/B against the round-5 library
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_prove.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2; do
  for v in staged direct; do echo "== TYPLONK_MSM_SCATTER=$v rep $rep"; TYPLONK_MSM_SCATTER=$v python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done
  echo "== r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  for v in staged direct; do echo "== batch TYPLONK_MSM_SCATTER=$v rep $rep"; TYPLONK_MSM_SCATTER=$v REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1; done
  echo "== batch r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
done > $O/msm_ab.txt 2>&1
cat $O/msm_ab.txt
REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline.txt 2>&1; cat $O/msm_timeline.txt
REPS=6 BATCH=9 rocprofv3 --kernel-trace --output-format csv -d $O/trace_b -- python3 tools/msm_batch_loop.py > $O/trace_b.log 2>&1
python3 - $(find $O/trace_b -name "*kernel_trace.csv" | head -1) > $O/batch_kernel_avgs.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0].replace("ty::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:50s} n={len(v):5d} avg={sum(v)/len(v):9.1f} us min={min(v):9.1f} total={sum(v)/1e3:9.2f} ms")
PY
cat $O/batch_kernel_avgs.txt
REPS=6 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/msm_loop.py > $O/pmc_write.log 2>&1
REPS=6 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/msm_loop.py > $O/pmc_fetch.log 2>&1
python3 tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > /dev/null; cat $O/pmc_traffic.json | head -80
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +2M -delete
for rep in 1 2; do for v in 2 1 0; do echo "== TYPLONK_PROVER_NTT_BATCH=$v rep $rep"; TYPLONK_PROVER_NTT_BATCH=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; echo "== r5base"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/prove_rounds.py 2>/dev/null | tail -2; done > $O/prove_ab.txt 2>&1
cat $O/prove_ab.txt
