#!/bin/bash
# round 6, call T: the two-launch reduction (ONE chain kernel behind the work-bound one instead of two) for the QUEUED MSMs of a
# batch / a proof: does an MSM that retires one accumulation earlier free its lane soon enough to matter?
export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O
for rep in 1 2 3; do for v in "none 10" "rc2 11"; do set -- $v
  echo "== REDUCE=$1 LOGW=$2 rep $rep"
  TYPLONK_MSM_REDUCE=$1 TYPLONK_MSM_RC2_LOGW=$2 REPS=10 python3 tools/msm_batch_loop.py 2>/dev/null | tail -n 1
  TYPLONK_MSM_REDUCE=$1 TYPLONK_MSM_RC2_LOGW=$2 python3 tools/prove_rounds.py 2>/dev/null | tail -n 2
done; done > $O/batch_prove.txt 2>&1; cat $O/batch_prove.txt
