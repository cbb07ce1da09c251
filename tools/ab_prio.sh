# same-box A/B: wave priority (s_setprio) in every kernel except the bucket accumulation, against a build without it;
# with / without the coset extensions at low priority; with / without serialised accumulations
mkdir -p gpurun_out/r2s
run() { echo "== $1"; python tools/prove_rounds.py 2>/dev/null | tail -2; }
for rep in 1 2; do
  TYPLONK_LIB_PATH=$PWD/tools/_ab/noprio/libtyplonk_hip.so run "noprio rep $rep"
  TYPLONK_LIB_PATH=$PWD/tools/_ab/noprio/libtyplonk_hip.so TYPLONK_MSM_SERIAL_ACC=1 run "noprio(no serial in that build) rep $rep"
  TYPLONK_NTT_EXT_PRIO=1 run "prio, extends high rep $rep"
  run "prio, extends low rep $rep"
  TYPLONK_MSM_SERIAL_ACC=1 run "prio, extends low, serial acc rep $rep"
  TYPLONK_MSM_SERIAL_ACC=1 TYPLONK_NTT_EXT_PRIO=1 run "prio, extends high, serial acc rep $rep"
done 2>&1 | tee gpurun_out/r2s/ab_prio.txt
