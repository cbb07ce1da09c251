# Round-2 profile collection (run on the GPU box through gpurun): kernel trace + stats of the bench command, PMC traffic
# of the MSM loop in separate passes (the guide's rule: one --pmc set per run, no other trace domains).
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_msm -- python3 bench.py --steps 20 --warmup 5 --msm-only > $O/bench_msm_only.json 2> $O/bench_msm.err
REPS=10 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/msm_loop.py > $O/pmc_fetch.log 2>&1
REPS=10 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/msm_loop.py > $O/pmc_write.log 2>&1
REPS=10 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/msm_loop.py > $O/pmc_sq.log 2>&1
SIZES=20,22 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ntt -- python3 tools/ntt_bench.py > $O/ntt.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" | head; du -sh $O
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $O/pmc_summary.json | head -40
