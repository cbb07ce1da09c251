#!/bin/bash
# round 6, call L: typlonk_prove_host -- parity, then against upload + typlonk_prove
export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_prove.py tests/test_gpu_robustness.py tests/test_gpu_prover_ops.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python3 tools/prove_host_path.py 2>/dev/null | grep PROVEHOST > $O/prove_host.txt; LOG_N=22 python3 tools/prove_host_path.py 2>/dev/null | grep PROVEHOST >> $O/prove_host.txt; cat $O/prove_host.txt
