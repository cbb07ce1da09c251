#!/bin/bash
# round 6, call M: bench.py with the timed region bracketing only the dominant kernel (typlonk_set_profiling 2)
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
for rep in 1 2 3; do python3 bench.py --msm-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['msm_stage_ms'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['msm_batch'], d['prove_native_ms'], d['ntt']['kernel_ms'], d['ntt']['batched']['kernel_ms_per_transform'], d['msm_stage_ms'], d['roofline']['frac'], d['roofline']['kernel_ms'], [k for k in d if k.endswith('_error')])"
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "bench" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
