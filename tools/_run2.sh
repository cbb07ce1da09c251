export TMPDIR=/tmp
O=gpurun_out/r03_b; mkdir -p $O
for cfg in "17 4 rc2" "20 1 rc2" "20 1 rc4"; do set -- $cfg
  TABLES=$1 TYPLONK_MSM_LANES=$2 TYPLONK_MSM_REDUCE=$3 NO_EXCHANGE=1 WORLD=8 REPS=60 rocprofv3 --kernel-trace --output-format csv -d $O/t_$1_$2_$3 -- python3 tools/shard_latency.py > $O/t_$1_$2_$3.log 2>&1
  F=$(find $O/t_$1_$2_$3 -name "*kernel_trace.csv" | head -1)
  echo "== tables $1 lanes $2 reduce $3" >> $O/timelines.txt
  python3 tools/msm_timeline.py $F 40 >> $O/timelines.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete
cat $O/timelines.txt
