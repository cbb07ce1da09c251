export TMPDIR=/tmp
O=gpurun_out/r03_e; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_quotient.py -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
(TYPLONK_NTT_FR30=2 timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q) > $O/pytest_fr30.log 2>&1; tail -3 $O/pytest_fr30.log
for r in 2 4 2 4; do for f in 1; do echo "== radix $r fr30 $f" >> $O/ntt.txt; TYPLONK_NTT_RADIX=$r TYPLONK_NTT_FR30=$f SIZES=14,16,18,19,20,22,24 timeout 300 python tools/ntt_bench.py 2>/dev/null | grep -v '"coset": true' | cut -c1-150 >> $O/ntt.txt; done; done
for r in 2 4; do echo "== radix $r fr30 2 (30-bit at every size)" >> $O/ntt.txt; TYPLONK_NTT_RADIX=$r TYPLONK_NTT_FR30=2 SIZES=20,22 timeout 300 python tools/ntt_bench.py 2>/dev/null | grep -v '"coset": true' | cut -c1-150 >> $O/ntt.txt; done
cat $O/ntt.txt
