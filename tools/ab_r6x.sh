#!/bin/bash
# round 6, call X: the prover's evaluation slots in pinned host memory (a fetch = one synchronisation, no staged copy) and r(zeta)
# read after round 3's commitments instead of before them; TYPLONK_PROVER_FETCH=0 is the previous behaviour
export TMPDIR=/tmp
O=gpurun_out/r6x; mkdir -p $O
timeout 1700 python3 -m pytest tests/test_gpu_prove.py tests/test_gpu_prover_ops.py tests/test_gpu_full_size_vs_cpu.py tests/test_gpu_dist.py tests/test_host_mirror.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
for rep in 1 2 3; do for v in 1 0; do
  echo "== PROVER_FETCH=$v rep $rep"
  TYPLONK_PROVER_FETCH=$v python3 tools/prove_rounds.py 2>/dev/null | tail -n 2
done; done > $O/prove.txt 2>&1; cat $O/prove.txt
for v in 1 0; do echo "== 2^22 PROVER_FETCH=$v"; LOG_N=22 TYPLONK_PROVER_FETCH=$v python3 tools/prove_rounds.py 2>/dev/null | tail -n 2; done > $O/prove22.txt 2>&1; cat $O/prove22.txt
