mkdir -p gpurun_out/r2f
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export TYPLONK_LIB_PATH=$PWD/typlonk_amd/libtyplonk_hip_base.so; else unset TYPLONK_LIB_PATH; fi
  echo "== $lib rep $rep"
  CHUNKS=1,2 python tools/msm_chunks.py 2>/dev/null | tee -a gpurun_out/r2f/msm_$lib.jsonl
  SIZES=20,22 python tools/ntt_bench.py 2>/dev/null | tee -a gpurun_out/r2f/ntt_$lib.jsonl
  REPS=4 python tools/prove_loop.py 2>/dev/null | tee -a gpurun_out/r2f/prove_$lib.txt
done; done
unset TYPLONK_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_quotient.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -3
