#!/bin/bash
# SURVEY 8(d)'s CPU legs as written, once, on the final tree: the reference-faithful MSM on all 2^20 terms (one core) and the
# schoolbook quotient products at 2^10, 2^11, 2^12 and 2^14 (one core) -- ~12 minutes of host time on the GPU box
export TMPDIR=/tmp
O=gpurun_out/r6cpu; mkdir -p $O
python3 bench.py --cpu-full --no-prove-2-22 --steps 10 --warmup 3 > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err
tail -c 3000 $O/bench_cpu_full.json
