# same-box A/B of compiler scheduling strategies (libraries built into tools/_ab/<name>/ by hand)
mkdir -p gpurun_out/r2s
for rep in 1 2; do
for lib in base iterilp_msm iterilp_all nomisched_msm; do
  if [ $lib = base ]; then unset TYPLONK_LIB_PATH; else export TYPLONK_LIB_PATH=$PWD/tools/_ab/$lib/libtyplonk_hip.so; fi
  echo "== $lib rep $rep"
  CHUNKS=2 REPS=40 python tools/msm_chunks.py 2>/dev/null | grep "^{" | cut -c1-220
  SIZES=20,22 python tools/ntt_bench.py 2>/dev/null | cut -c1-150
  python tools/prove_rounds.py 2>/dev/null | tail -2
done; done 2>&1 | tee gpurun_out/r2s/ab_sched.txt
unset TYPLONK_LIB_PATH
