#!/bin/bash
# same-box A/B: the accumulation kernel with / without the peeled affine + affine second addition
# (tools/_ab/nopeel = python -m typlonk_amd.build variant nopeel msm_accum.hip:-DMSM_NO_PEEL)
for rep in 1 2; do
  for v in base nopeel; do
    if [ $v = base ]; then unset TYPLONK_LIB_PATH; else export TYPLONK_LIB_PATH=$PWD/tools/_ab/$v/libtyplonk_hip.so; fi
    echo "== $v rep $rep"
    CHUNKS=0,1 python3 tools/msm_chunks.py 2>/dev/null | grep log_m
    LOG_M=20 python3 tools/msm_batch_loop.py 2>/dev/null | grep BATCH
    WORLD=8 TABLES=auto REPS=40 python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" | python3 -c "import sys,json; [print({k:d[k] for k in ('local_msm_wall_ms','stages_ms','sharded_batch9_ms_per_msm')}) for d in (json.loads(l[6:]) for l in sys.stdin)]"
    python3 tools/prove_rounds.py 2>/dev/null | tail -2
  done
done
