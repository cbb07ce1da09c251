O=gpurun_out/r03_n; mkdir -p $O
for cfg in "4 17" "4 20" "2 17" "2 20" "4 17" "4 20"; do set -- $cfg; TABLES=$2 NO_EXCHANGE=1 WORLD=$1 REPS=200 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()[6:]); print('world',$1,'tables',$2,d['local_msm_wall_noprof_ms'],d['stages_ms'])"; done | tee $O/c17_vs_c20.txt
