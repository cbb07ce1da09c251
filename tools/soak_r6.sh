#!/bin/bash
# Round-6 soak on the final tree: the two differential fuzzers (tools/fuzz_gpu.py now also covers groups of transforms and
# every form of the bucket sort; tools/fuzz_prove.py also the host-column proof).  Output under gpurun_out/r6soak/.
mkdir -p gpurun_out/r6soak
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6soak/build.log 2>&1
FUZZ_SECONDS=${GPU_FUZZ_SECONDS:-1200} FUZZ_SEED=${GPU_FUZZ_SEED:-6061} FUZZ_MAX_LOG=22 timeout 1500 python tools/fuzz_gpu.py > gpurun_out/r6soak/fuzz_gpu.txt 2>&1
echo "fuzz_gpu rc=$?" >> gpurun_out/r6soak/fuzz_gpu.txt
FUZZ_SECONDS=${PROVE_FUZZ_SECONDS:-720} FUZZ_SEED=${PROVE_FUZZ_SEED:-66} FUZZ_MAX_LOG=18 timeout 1000 python tools/fuzz_prove.py > gpurun_out/r6soak/fuzz_prove.txt 2>&1
echo "fuzz_prove rc=$?" >> gpurun_out/r6soak/fuzz_prove.txt
tail -n 3 gpurun_out/r6soak/fuzz_gpu.txt; tail -n 3 gpurun_out/r6soak/fuzz_prove.txt
