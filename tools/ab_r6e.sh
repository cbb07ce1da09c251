#!/bin/bash
# round 6, call E: NTT with nine-limb inter-pass data (TYPLONK_NTT_L9) -- parity, same-box A/B, SQ counters; sort re-check
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_gpu_quotient.py tests/test_gpu_prove.py tests/test_gpu_prover_ops.py tests/test_gpu_msm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2; do for v in 1 0; do echo "== TYPLONK_NTT_L9=$v rep $rep"; TYPLONK_NTT_L9=$v SIZES=16,18,20,22,24 python3 tools/ntt_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['log_n'], 'inv' if d['inverse'] else 'fwd', 'coset' if d['coset'] else '     ', d['kernel_ms'], d['passes'])"; done; done > $O/ntt_ab.txt 2>&1
cat $O/ntt_ab.txt
for v in 1 0; do echo "== batch TYPLONK_NTT_L9=$v"; TYPLONK_NTT_L9=$v SIZES=20 COUNTS=3 python3 tools/ntt_batch_bench.py 2>/dev/null; TYPLONK_NTT_BIG=0 TYPLONK_NTT_L9=$v SIZES=20 COUNTS=3 python3 tools/ntt_batch_bench.py 2>/dev/null; done > $O/ntt_batch_ab.txt 2>&1; cat $O/ntt_batch_ab.txt
for v in 1 0; do
TYPLONK_NTT_L9=$v SIZES=20,22 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/ntt_sq_$v -- python3 tools/ntt_bench.py > $O/nsq.log 2>&1
python3 tools/pmc_sq_summary.py $O/ntt_sq_$v > $O/pmc_sq_ntt_l9_$v.json 2>/dev/null; done
python3 -c "
import json
for v in (1,0):
    d=json.load(open('$O/pmc_sq_ntt_l9_%d.json'%v))
    for k,x in d.items():
        if 'ntt_pass' in k: print(v, k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in x.items()})"
find $O -name "*counter_collection.csv" -size +2M -delete
for rep in 1 2; do for v in 1 0; do echo "== prove TYPLONK_NTT_L9=$v"; TYPLONK_NTT_L9=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done; done > $O/prove_ab.txt 2>&1; cat $O/prove_ab.txt
for v in 1 0; do echo "== quotient TYPLONK_NTT_L9=$v"; TYPLONK_NTT_L9=$v python3 tools/quotient_loop.py 2>/dev/null | tail -2; done > $O/quot_ab.txt 2>&1; cat $O/quot_ab.txt
REPS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/msm_loop.py > $O/trace.log 2>&1
python3 tools/msm_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 10 > $O/msm_timeline.txt 2>&1; cat $O/msm_timeline.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
