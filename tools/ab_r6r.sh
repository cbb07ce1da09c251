#!/bin/bash
# round 6, call R: bench.py as the FIRST program on a fresh box, with and without the clock-ramp warm-up; the N = 2 launch path
export TMPDIR=/tmp
O=gpurun_out/r6r; mkdir -p $O
python3 bench.py --msm-only --warm-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cold, no ramp', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('warm_extra_steps'))"
sleep 20
python3 bench.py --msm-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('after 20 s idle, ramp 1 s', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('warm_extra_steps'))"
sleep 20
python3 bench.py --msm-only --warm-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('after 20 s idle, no ramp', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('warm_extra_steps'))"
sleep 20
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['msm_batch'], d['prove_native_ms'], d['warm_extra_steps'], d['roofline']['kernel_ms'], [k for k in d if k.endswith('_error')])"
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "bench" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
