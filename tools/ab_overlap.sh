mkdir -p gpurun_out/r2h
for rep in 1 2; do for ov in 0 1 3 5 7 4; do
  echo "== TYPLONK_PROVER_OVERLAP=$ov rep $rep"
  TYPLONK_PROVER_OVERLAP=$ov python tools/prove_rounds.py 2>/dev/null | tail -2
done; done | tee gpurun_out/r2h/overlap.txt
