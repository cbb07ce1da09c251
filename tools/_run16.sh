O=gpurun_out/r03_p; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_quotient.py tests/test_gpu_prove.py -x -q 2>&1 | tail -3)
(TYPLONK_NTT_FR30=2 timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_quotient.py -x -q 2>&1 | tail -2)
(TYPLONK_NTT_FR30=0 TYPLONK_NTT_BIG=0 timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q 2>&1 | tail -2)
for d in 0 1 0 1; do echo "== direct $d" >> $O/ntt.txt; TYPLONK_NTT_DIRECT=$d SIZES=14,16,18,19,20,22,24 timeout 300 python tools/ntt_bench.py 2>/dev/null | grep -v '"coset": true' | cut -c1-120 >> $O/ntt.txt; done
cat $O/ntt.txt
