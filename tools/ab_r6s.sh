#!/bin/bash
# round 6, call S: share of the terms in the FIRST chunk of a stand-alone MSM (its sort is the exposed one)
export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
TYPLONK_MSM_FIRST_PCT=25 timeout 900 python3 -m pytest tests/test_gpu_msm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
for rep in 1 2 3; do for p in 0 25 33 40 15; do echo "== 2^20 FIRST_PCT=$p rep $rep"; TYPLONK_MSM_FIRST_PCT=$p python3 bench.py --msm-only --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done; done > $O/first.txt 2>&1; cat $O/first.txt
for rep in 1 2; do for p in 0 6 9; do echo "== 2^22 FIRST_PCT=$p rep $rep"; TYPLONK_MSM_FIRST_PCT=$p python3 bench.py --log-n 22 --msm-only --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done; done > $O/m22.txt 2>&1; cat $O/m22.txt
# the two-launch reduction at 2^19 buckets with 2^10 / 2^11 / 2^12 wavefronts in its first launch, against the four-launch form
TYPLONK_MSM_REDUCE=rc2 TYPLONK_MSM_RC2_LOGW=11 timeout 900 python3 -m pytest tests/test_gpu_msm.py -x -q -m gpu > $O/pytest_rc2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rc2.log; tail -n 3 $O/pytest_rc2.log
for rep in 1 2 3; do for v in "none 10" "rc2 10" "rc2 11" "rc2 12"; do set -- $v; echo "== 2^20 REDUCE=$1 LOGW=$2 rep $rep"; TYPLONK_MSM_REDUCE=$1 TYPLONK_MSM_RC2_LOGW=$2 python3 bench.py --msm-only --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; done; done > $O/rc2.txt 2>&1; cat $O/rc2.txt
