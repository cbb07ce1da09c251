#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
LOG_N=22 TABLES=auto timeout 1200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29555 tests/dist_prove_worker.py > $O/w22.log 2>&1; echo "rc=$?" >> $O/w22.log
grep -n "Error\|error\|Traceback\|typlonk\|rc=" $O/w22.log | head -40
rocm-smi --showmeminfo vram 2>/dev/null | tail -5
