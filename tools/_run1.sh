mkdir -p gpurun_out/r03_a
(timeout 900 python -m pytest tests/test_gpu_msm_shard.py tests/test_gpu_msm.py -x -q) > gpurun_out/r03_a/pytest.log 2>&1; tail -5 gpurun_out/r03_a/pytest.log
for cfg in "20 1 rc4" "20 1 rc2" "17 1 rc2" "17 2 rc2" "17 4 rc2" "16 4 rc2" "16 8 rc2" "15 4 rc2" "17 2 rc4"; do set -- $cfg; TABLES=$1 TYPLONK_MSM_LANES=$2 TYPLONK_MSM_REDUCE=$3 NO_EXCHANGE=1 WORLD=8 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" >> gpurun_out/r03_a/shard.jsonl; done
for w in 2 4; do TABLES=auto NO_EXCHANGE=1 WORLD=$w timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" >> gpurun_out/r03_a/shard.jsonl; TABLES=20 TYPLONK_MSM_LANES=1 NO_EXCHANGE=1 WORLD=$w timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" >> gpurun_out/r03_a/shard.jsonl; done
cat gpurun_out/r03_a/shard.jsonl
