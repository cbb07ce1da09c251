#!/bin/bash
# final validation of the round-6 tree: the whole GPU suite, smoke, the default bench line
export TMPDIR=/tmp
O=gpurun_out/r6final; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
strings typlonk_amd/libtyplonk_hip.so | grep -c TYPLONK_TEST
