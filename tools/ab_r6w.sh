#!/bin/bash
# round 6, call W: host-scalar commitments BELOW 2^20 terms (one chunk by default: the whole copy is exposed): chunks for this path?
export TMPDIR=/tmp
O=gpurun_out/r6w; mkdir -p $O
for rep in 1 2; do for v in "0 0" "0 2" "0 3" "25 2" "25 3" "33 3"; do set -- $v
  echo "== FIRST_PCT=$1 CHUNKS=$2 rep $rep"
  TYPLONK_MSM_FIRST_PCT=$1 TYPLONK_MSM_CHUNKS=$2 SIZES=17,18,19 python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH
done; done > $O/hostpath.txt 2>&1; cat $O/hostpath.txt
