export TMPDIR=/tmp
O=gpurun_out/r03_g; mkdir -p $O
REPS=40 rocprofv3 --kernel-trace --output-format csv -d $O/t20 -- python3 tools/msm_loop.py > $O/t20.log 2>&1
python3 tools/msm_timeline.py $(find $O/t20 -name "*kernel_trace.csv" | head -1) 30 > $O/timeline_2_20.txt 2>&1; cat $O/timeline_2_20.txt
find $O -name "*kernel_trace.csv" -delete
for ch in 1 2; do TYPLONK_MSM_CHUNKS=$ch python bench.py --steps 20 --warmup 5 --msm-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunks',$ch,d['ms_per_step'],d['msm_stage_ms'])"; done
