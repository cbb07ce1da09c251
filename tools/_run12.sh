O=gpurun_out/r03_l; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_msm_shard.py -x -q 2>&1 | tail -2)
for sp in 0 1 0 1; do TYPLONK_MSM_LANES_SPLIT=$sp NO_EXCHANGE=1 WORLD=8 REPS=300 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()[6:]); print('split',$sp,d['local_msm_wall_noprof_ms'],d['stages_ms'])"; done | tee $O/split.txt
