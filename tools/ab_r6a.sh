#!/bin/bash
# round 6, call A: parity of the new entry points, then same-box measurements
export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ntt.py tests/test_gpu_prover_ops.py tests/test_gpu_prove.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for big in 0 2; do TYPLONK_NTT_BIG=$big SIZES=20 COUNTS=1,3,5 python3 tools/ntt_batch_bench.py 2>/dev/null >> $O/ntt_batch.jsonl; done
SIZES=16,18,22 COUNTS=3 python3 tools/ntt_batch_bench.py 2>/dev/null >> $O/ntt_batch.jsonl
cat $O/ntt_batch.jsonl
for rep in 1 2 3; do
  for v in 1 0; do echo "== TYPLONK_PROVER_NTT_BATCH=$v rep $rep"; TYPLONK_PROVER_NTT_BATCH=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done
  echo "== r5base rep $rep"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/prove_rounds.py 2>/dev/null | tail -2
done > $O/prove_ab.txt 2>&1
cat $O/prove_ab.txt
