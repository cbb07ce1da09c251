#!/bin/bash
# round 5, first GPU batch: new parity tests, inversion pricing + affine micro-kernel, set-up cost
mkdir -p gpurun_out
python -m pytest tests/test_gpu_msm.py -x -q -m gpu -k "inversion or identity_bases or srs_generate or fixed_base" 2>&1 | tail -15 > gpurun_out/r5a_tests.txt
cat gpurun_out/r5a_tests.txt
timeout 600 ./tools/ubench5 > gpurun_out/r05_ubench_inversion.txt 2>&1
cat gpurun_out/r05_ubench_inversion.txt
timeout 600 python tools/srs_setup_bench.py > gpurun_out/r05_srs_setup.jsonl 2>&1
cat gpurun_out/r05_srs_setup.jsonl
