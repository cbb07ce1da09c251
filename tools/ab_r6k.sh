#!/bin/bash
# round 6, call K: host-scalar path with the copy chunked beside the kernels; quotient with batched extensions
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_quotient.py tests/test_gpu_msm_shard.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do echo "== new"; python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH; echo "== r5base"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH; done > $O/hostpath.txt 2>&1; cat $O/hostpath.txt
for rep in 1 2; do echo "== quotient new"; python3 tools/quotient_loop.py 2>/dev/null | tail -1; echo "== quotient r5base"; TYPLONK_LIB_PATH=tools/_ab/r5base/libtyplonk_hip.so python3 tools/quotient_loop.py 2>/dev/null | tail -1; done > $O/quot.txt 2>&1; cat $O/quot.txt
