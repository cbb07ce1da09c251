#!/bin/bash
mkdir -p gpurun_out
timeout 600 ./tools/ubench5 > gpurun_out/r05_ubench_inversion.txt 2>&1
cat gpurun_out/r05_ubench_inversion.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5b_tests.txt
cat gpurun_out/r5b_tests.txt
