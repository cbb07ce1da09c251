#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do python tools/prove_rounds.py 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r5c_rounds.txt
cat gpurun_out/r5c_rounds.txt
python -m pytest tests/test_gpu_prove.py tests/test_gpu_full_size_vs_cpu.py -x -q -m gpu 2>&1 | tail -5
