export TMPDIR=/tmp
O=gpurun_out/r03_h; mkdir -p $O
for st in 0 1 0 1; do TYPLONK_MSM_STAGGER=$st python bench.py --steps 30 --warmup 8 --msm-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger',$st,round(d['ms_per_step'],4),d['msm_stage_ms'])"; done 2>&1 | tee $O/stagger.txt
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
