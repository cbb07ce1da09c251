# round 4, run B: parity suite after the host-driver split, the pricing micro-benchmark, the driver's bench invocation
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
./tools/ubench4 > $O/ubench4.txt 2>&1; cat $O/ubench4.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
python3 tools/prove_rounds.py > $O/prove_rounds.txt 2>&1; tail -8 $O/prove_rounds.txt
