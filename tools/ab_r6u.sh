#!/bin/bash
# round 6, call U: typlonk_msm_g1 from HOST scalars -- the first chunk's copy is the exposed one (16.8 MB at 2^20): smaller first
# chunks / more chunks for this path only?
export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O
for rep in 1 2; do for v in "0 0" "0 3" "0 4" "12 2" "12 3" "25 2" "25 3" "6 3" "6 4"; do set -- $v
  echo "== FIRST_PCT=$1 CHUNKS=$2 rep $rep"
  TYPLONK_MSM_FIRST_PCT=$1 TYPLONK_MSM_CHUNKS=$2 SIZES=20 python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH
done; done > $O/hostpath.txt 2>&1; cat $O/hostpath.txt
