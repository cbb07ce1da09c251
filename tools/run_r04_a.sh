# round 4, run A: parity suite, the driver's bench invocation, the long CPU legs on record, a baseline proof timeline
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --cpu-full > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err
GAP_MS=8 REPS=4 rocprofv3 --kernel-trace --output-format csv -d $O/prove_trace -- python3 tools/prove_loop.py > $O/prove_loop.log 2>&1
python3 tools/trace_timeline.py $(find $O/prove_trace -name "*kernel_trace.csv" | head -1) > $O/prove_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
tail -3 $O/pytest.log; tail -c 600 $O/bench.json; tail -2 $O/prove_loop.log; tail -1 $O/prove_timeline.txt
