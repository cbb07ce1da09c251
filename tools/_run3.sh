export TMPDIR=/tmp
O=gpurun_out/r03_c; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_msm_shard.py tests/test_gpu_msm.py -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "17 4 rc2 8" "17 2 rc2 8" "auto 0 rc2 4" "auto 0 rc2 2" "20 1 rc4 8"; do set -- $cfg
  L=$2; [ "$L" = "0" ] && unset LANES_ENV || LANES_ENV=$L
  if [ "$L" = "0" ]; then TABLES=$1 TYPLONK_MSM_REDUCE=$3 NO_EXCHANGE=1 WORLD=$4 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" >> $O/shard.jsonl
  else TABLES=$1 TYPLONK_MSM_LANES=$L TYPLONK_MSM_REDUCE=$3 NO_EXCHANGE=1 WORLD=$4 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" >> $O/shard.jsonl; fi
done
cat $O/shard.jsonl
TABLES=17 TYPLONK_MSM_LANES=4 NO_EXCHANGE=1 WORLD=8 REPS=60 rocprofv3 --kernel-trace --output-format csv -d $O/t17 -- python3 tools/shard_latency.py > $O/t17.log 2>&1
python3 tools/msm_timeline.py $(find $O/t17 -name "*kernel_trace.csv" | head -1) 40 > $O/timeline_17_4.txt 2>&1; cat $O/timeline_17_4.txt
REPS=40 rocprofv3 --kernel-trace --output-format csv -d $O/t20 -- python3 tools/msm_loop.py > $O/t20.log 2>&1
python3 tools/msm_timeline.py $(find $O/t20 -name "*kernel_trace.csv" | head -1) 30 > $O/timeline_2_20.txt 2>&1; cat $O/timeline_2_20.txt
find $O -name "*kernel_trace.csv" -delete
python bench.py --steps 20 --warmup 5 --msm-only 2>/dev/null | tail -1 > $O/bench_msm_only.json; cat $O/bench_msm_only.json | cut -c1-900
