O=gpurun_out/r03_j; mkdir -p $O
T="tests/test_gpu_msm.py tests/test_gpu_msm_shard.py tests/test_gpu_prove.py tests/test_gpu_prover_ops.py tests/test_gpu_ntt.py tests/test_gpu_quotient.py"
run() { name=$1; shift; (env "$@" timeout 1500 python -m pytest $T -x -q 2>&1 | tail -2) > $O/$name.log; echo "$name: $(tail -1 $O/$name.log)"; }
run inflight1 TYPLONK_MSM_INFLIGHT=1
run inflight2_overlap7 TYPLONK_MSM_INFLIGHT=2 TYPLONK_PROVER_OVERLAP=7
run chunks3_overlap0 TYPLONK_MSM_CHUNKS=3 TYPLONK_PROVER_OVERLAP=0 TYPLONK_MSM_STAGGER=0
run running_nobig TYPLONK_MSM_REDUCE=running TYPLONK_NTT_BIG=0
run rc2_forced_lanes4 TYPLONK_MSM_REDUCE=rc2 TYPLONK_MSM_LANES=4
run scan3_ordersplit_radix2 TYPLONK_MSM_SCAN=scan3 TYPLONK_MSM_ORDER=split TYPLONK_NTT_RADIX=2
run nofulltables_fr30_2 TYPLONK_NTT_FULL_TABLES=0 TYPLONK_NTT_FR30=2
run atomic_sort TYPLONK_MSM_SORT=atomic
