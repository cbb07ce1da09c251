#!/bin/bash
# round 6, call O: sorts beside an accumulation with 256-thread level 1 + raised wavefront priority (TYPLONK_MSM_SORT_PRIO)
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_prove.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for v in 1 0; do
TYPLONK_MSM_SORT_PRIO=$v REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace$v -- python3 tools/msm_loop.py > $O/trace.log 2>&1
echo "== TYPLONK_MSM_SORT_PRIO=$v"; python3 tools/msm_timeline.py $(find $O/trace$v -name "*kernel_trace.csv" | head -1) 10 2>&1 | tee $O/msm_timeline_$v.txt
done
find $O -name "*kernel_trace.csv" -size +4M -delete
for rep in 1 2 3; do for v in 1 0; do echo "== PRIO=$v rep $rep"; TYPLONK_MSM_SORT_PRIO=$v python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'])"; echo "== batch PRIO=$v"; TYPLONK_MSM_SORT_PRIO=$v REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -1; done; done > $O/ab.txt 2>&1; cat $O/ab.txt
for rep in 1 2 3; do for v in 1 0; do echo "== prove PRIO=$v"; TYPLONK_MSM_SORT_PRIO=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; done > $O/prove_ab.txt 2>&1; cat $O/prove_ab.txt
for v in 1 0; do echo "== 2^22 prove PRIO=$v"; LOG_N=22 TYPLONK_MSM_SORT_PRIO=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done > $O/prove_ab_22.txt 2>&1; cat $O/prove_ab_22.txt
