O=gpurun_out/r03_i; mkdir -p $O
for w in 8 4 2; do WORLD=$w REPS=200 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" | tail -1 >> $O/shard_latency.jsonl; done
WORLD=8 TABLES=20 TYPLONK_MSM_LANES=1 TYPLONK_MSM_REDUCE=rc4 REPS=200 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" | tail -1 >> $O/shard_latency.jsonl
cat $O/shard_latency.jsonl
