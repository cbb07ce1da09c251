O=gpurun_out/r03_k; mkdir -p $O
for p in 0 1 0 1; do TYPLONK_MSM_SIDE_PRIO=$p python bench.py --steps 30 --warmup 8 --msm-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side_prio',$p,round(d['ms_per_step'],4),{k:round(v,3) for k,v in d['msm_stage_ms'].items()})"; done 2>&1 | tee $O/prio.txt
for w in 8; do WORLD=$w REPS=200 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" | tail -1; done | tee -a $O/prio.txt
