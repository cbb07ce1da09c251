export TMPDIR=/tmp
O=gpurun_out/r4x; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
