#!/bin/bash
# round 6, call V: the host-scalar path with its short first chunk (default now) against equal chunks (TYPLONK_MSM_CHUNKS forces them)
export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_full_size_vs_cpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
for rep in 1 2 3; do
  echo "== short first chunk (default) rep $rep"; python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH
  echo "== equal chunks (2 at 2^20, 8 at 2^22) rep $rep"; SIZES=20 TYPLONK_MSM_CHUNKS=2 python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH; SIZES=22 TYPLONK_MSM_CHUNKS=8 python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH
done > $O/hostpath.txt 2>&1; cat $O/hostpath.txt
