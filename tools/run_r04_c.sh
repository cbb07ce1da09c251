# round 4, run C: SIMD-aware pricing micro-benchmark + same-box A/B of the accumulation kernel's register budget
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
./tools/ubench4 > $O/ubench4.txt 2>&1; cat $O/ubench4.txt
for rep in 1 2; do
for lib in base w3 w4; do
  if [ $lib = base ]; then unset TYPLONK_LIB_PATH; else export TYPLONK_LIB_PATH=$PWD/tools/_ab/$lib/libtyplonk_hip.so; fi
  echo "== $lib rep $rep"
  CHUNKS=2 python tools/msm_chunks.py 2>/dev/null | grep "^{" | cut -c1-260
  python tools/prove_rounds.py 2>/dev/null | tail -2
done; done 2>&1 | tee $O/ab_waves.txt
unset TYPLONK_LIB_PATH
