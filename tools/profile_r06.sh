# Round-6 profile collection (run on the GPU box through gpurun): the driver's bench invocation, kernel trace + stats of the
# MSM-only bench command, PMC traffic and SQ counters of the MSM loop in separate passes (the guide's rule: one --pmc set per
# run, no other trace domains), NTT stats single and batched, the shard-sized MSM, a proof timeline, the 2^22 configuration.
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_msm -- python3 bench.py --steps 20 --warmup 5 --msm-only > $O/bench_msm_only.json 2> $O/bench_msm.err
python3 tools/msm_timeline.py $(find $O/bench_msm -name "*kernel_trace.csv" | head -1) 15 > $O/msm_2_20_timeline.txt 2>&1
REPS=10 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/msm_loop.py > $O/pmc_fetch.log 2>&1
REPS=10 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/msm_loop.py > $O/pmc_write.log 2>&1
REPS=10 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/msm_loop.py > $O/pmc_sq.log 2>&1
SIZES=20,22 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ntt -- python3 tools/ntt_bench.py > $O/ntt.log 2>&1
SIZES=20,22 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ntt_fetch -- python3 tools/ntt_bench.py > $O/nf.log 2>&1
SIZES=20,22 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ntt_write -- python3 tools/ntt_bench.py > $O/nw.log 2>&1
SIZES=20,22 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/ntt_sq -- python3 tools/ntt_bench.py > $O/nsq.log 2>&1
python3 tools/pmc_sq_summary.py $O/ntt_sq > $O/pmc_sq_ntt.json 2>/dev/null
SIZES=20 COUNTS=3 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/nttb_sq -- python3 tools/ntt_batch_bench.py > $O/nbsq.log 2>&1
python3 tools/pmc_sq_summary.py $O/nttb_sq > $O/pmc_sq_ntt_batched.json 2>/dev/null
for big in 0 2; do TYPLONK_NTT_BIG=$big SIZES=20 COUNTS=1,3,5 python3 tools/ntt_batch_bench.py 2>/dev/null >> $O/ntt_batch.jsonl; done
SIZES=16,18,22 COUNTS=3 python3 tools/ntt_batch_bench.py 2>/dev/null >> $O/ntt_batch.jsonl
python3 - "$(find $O/ntt_fetch -name '*counter_collection.csv' | head -1)" "$(find $O/ntt_write -name '*counter_collection.csv' | head -1)" $O/ntt_pmc.json <<'PY'
import csv, sys, collections, json
def per(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("ty::ntt_pass"):
            agg[(r["Kernel_Name"].split("(")[0].replace("ty::", ""), int(r["Grid_Size"]), int(r.get("Workgroup_Size", 0) or 0))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f, w = per(sys.argv[1], "FETCH_SIZE"), per(sys.argv[2], "WRITE_SIZE")
out = {}
for (name, grid, wg) in sorted(f):
    n = grid * 4  # radix-4 groups: one thread per four elements of the tile
    out[f"{name} n=2^{n.bit_length() - 1}" + ("" if f"{name} n=2^{n.bit_length() - 1}" not in out else f" wg{wg}")] = {
        "grid": grid, "workgroup": wg, "fetch_kib_raw": f[(name, grid, wg)], "write_kib": w.get((name, grid, wg), 0.0),
        "traffic_bytes_per_pass": (2 * f[(name, grid, wg)] + w.get((name, grid, wg), 0.0)) * 1024, "algorithmic_bytes_per_transform": 64 * n}
open(sys.argv[3], "w").write(json.dumps(out, indent=1) + "\n")
PY
REPS=3 BATCH=9 rocprofv3 --kernel-trace --output-format csv -d $O/batch_trace -- python3 tools/msm_batch_loop.py > $O/batch_loop.log 2>&1
python3 tools/trace_timeline.py $(find $O/batch_trace -name "*kernel_trace.csv" | head -1) > $O/msm_batch_timeline.txt 2>&1
for W in 2 4 8; do TABLES=auto WORLD=$W REPS=60 python3 tools/shard_latency.py 2>/dev/null | grep "^SHARD" >> $O/shard_latency.jsonl; done
GAP_MS=8 REPS=4 rocprofv3 --kernel-trace --output-format csv -d $O/prove_trace -- python3 tools/prove_loop.py > $O/prove_loop.log 2>&1
python3 tools/prove_gaps.py $(find $O/prove_trace -name "*kernel_trace.csv" | head -1) > $O/prove_timeline.txt 2>&1
python3 tools/prove_rounds.py > $O/prove_rounds.txt 2>&1
LOG_N=22 python3 tools/prove_rounds.py 2>/dev/null | tail -2 > $O/prove_rounds_2_22.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $O/pmc_summary.json > /dev/null
python3 tools/pmc_sq_summary.py $O/pmc_sq > $O/pmc_sq_msm.json 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete
python3 bench.py --log-n 22 --cpu-sample 8192 --steps 10 --warmup 3 > $O/bench_2_22.json 2> $O/bench_2_22.err
python3 tools/srs_setup_bench.py > $O/srs_setup.jsonl 2>/dev/null
ls -la $O | head -50; tail -c 1500 $O/bench_full.json; cat $O/shard_latency.jsonl; tail -25 $O/prove_timeline.txt; cat $O/prove_rounds.txt | tail -3; cat $O/prove_rounds_2_22.txt; cat $O/msm_2_20_timeline.txt; cat $O/pmc_sq_ntt_batched.json
# the host-memory seams (the reference's commit() and prove() as a host caller sees them)
python3 tools/msm_host_path.py 2>/dev/null | grep HOSTPATH > $O/host_scalar_path.txt
python3 tools/prove_host_path.py 2>/dev/null | grep PROVEHOST > $O/prove_host_path.txt
