#!/bin/bash
# round 6, call Q: chunk count of a stand-alone MSM now that overlapped sorts are short; 2^22 stand-alone with / without sort priority
export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_msm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do for c in 0 2 3 4; do echo "== 2^20 CHUNKS=$c rep $rep"; TYPLONK_MSM_CHUNKS=$c python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done; done > $O/chunks.txt 2>&1; cat $O/chunks.txt
for rep in 1 2; do for v in 1 0; do echo "== 2^22 PRIO=$v rep $rep"; TYPLONK_MSM_SORT_PRIO=$v python3 bench.py --log-n 22 --msm-only --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done; done > $O/m22.txt 2>&1; cat $O/m22.txt
for c in 4 6 8; do echo "== 2^22 CHUNKS=$c"; TYPLONK_MSM_CHUNKS=$c python3 bench.py --log-n 22 --msm-only --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; done >> $O/m22.txt 2>&1; tail -6 $O/m22.txt
