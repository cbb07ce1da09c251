mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export TYPLONK_LIB_PATH=$PWD/typlonk_amd/libtyplonk_hip_base.so; else unset TYPLONK_LIB_PATH; fi
  echo "== $lib rep $rep"
  SIZES=20,22 python tools/ntt_bench.py 2>/dev/null | tee -a gpurun_out/r2i/ntt_$lib.jsonl | cut -c1-150
  python tools/prove_rounds.py 2>/dev/null | tail -2 | tee -a gpurun_out/r2i/prove_$lib.txt
done; done
unset TYPLONK_LIB_PATH
