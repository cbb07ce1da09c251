#!/bin/bash
# same-box A/B of the round-5 queueing fixes in prove(): TYPLONK_PROVER_PIPE = 0 as rounds 1-4 queued the commitments,
# 1 = round 3 behind one fence (shipped)
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== TYPLONK_PROVER_PIPE=$v rep $rep"
    TYPLONK_PROVER_PIPE=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2
  done
done
for v in 1 0; do echo "== 2^22 TYPLONK_PROVER_PIPE=$v"; LOG_N=22 TYPLONK_PROVER_PIPE=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done
for v in 1 0; do echo "== 2^16 TYPLONK_PROVER_PIPE=$v"; LOG_N=16 TABLES=0 TYPLONK_PROVER_PIPE=$v python3 tools/prove_rounds.py 2>/dev/null | tail -2; done
