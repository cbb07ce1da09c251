#!/bin/bash
# round 6, call C: sort polish + column-parallel multiplication in the reduction chains -- parity, then A/B vs round 5
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_prove.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2 3; do
  echo "== new rep $rep"; python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"
  echo "== batch new rep $rep"; REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
  echo "== batch r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1
done > $O/msm_ab.txt 2>&1
cat $O/msm_ab.txt
REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline.txt 2>&1; cat $O/msm_timeline.txt
REPS=6 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/msm_loop.py > $O/pmc_write.log 2>&1
REPS=6 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/msm_loop.py > $O/pmc_fetch.log 2>&1
python3 tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > /dev/null; python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json'))
for k,v in d.items(): print(k, v['launches'], 'fetch KiB', round(v['fetch_size_kib_raw']), 'write KiB', round(v['write_size_kib']))"
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +2M -delete
for rep in 1 2; do echo "== new"; python3 tools/prove_rounds.py 2>/dev/null | tail -2; echo "== r5base"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/prove_rounds.py 2>/dev/null | tail -2; done > $O/prove_ab.txt 2>&1
cat $O/prove_ab.txt
WORLD=8 TABLES=auto REPS=40 python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" > $O/shard8.txt; WORLD=8 TABLES=auto REPS=40 TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" >> $O/shard8.txt; cat $O/shard8.txt
