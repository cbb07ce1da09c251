# Round-3 profile collection (run on the GPU box through gpurun): kernel trace + stats of the bench command, PMC traffic
# of the MSM loop and of the NTT passes in separate passes (the guide's rule: one --pmc set per run, no other trace
# domains), SQ counters, and the same for the shard-sized MSM (an 8-way index shard of the 2^20 commitment).
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_msm -- python3 bench.py --steps 20 --warmup 5 --msm-only > $O/bench_msm_only.json 2> $O/bench_msm.err
REPS=10 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/msm_loop.py > $O/pmc_fetch.log 2>&1
REPS=10 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/msm_loop.py > $O/pmc_write.log 2>&1
REPS=10 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/msm_loop.py > $O/pmc_sq.log 2>&1
SIZES=20,22 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ntt -- python3 tools/ntt_bench.py > $O/ntt.log 2>&1
SIZES=20,22 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ntt_fetch -- python3 tools/ntt_bench.py > $O/nf.log 2>&1
SIZES=20,22 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ntt_write -- python3 tools/ntt_bench.py > $O/nw.log 2>&1
SIZES=20,22 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/ntt_sq -- python3 tools/ntt_bench.py > $O/nsq.log 2>&1
TABLES=auto NO_EXCHANGE=1 WORLD=8 REPS=60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard -- python3 tools/shard_latency.py > $O/shard.log 2>&1
python3 tools/msm_timeline.py $(find $O/shard -name "*kernel_trace.csv" | head -1) 40 > $O/shard_timeline.txt 2>&1
python3 tools/msm_timeline.py $(find $O/bench_msm -name "*kernel_trace.csv" | head -1) 15 > $O/msm_2_20_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $O/pmc_summary.json > /dev/null
python3 tools/pmc_sq_summary.py $O/pmc_sq > $O/pmc_sq_msm.json 2>/dev/null
python3 tools/pmc_sq_summary.py $O/ntt_sq > $O/pmc_sq_ntt.json 2>/dev/null
F=$(find $O/ntt_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/ntt_write -name "*counter_collection.csv" | head -1)
python3 - "$F" "$W" <<'PY'
import csv, sys, collections, json
def per(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("ty::ntt_pass_kernel"):
            agg[(int(r["Grid_Size"]), int(r.get("Workgroup_Size", 0) or 0))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f, w = per(sys.argv[1], "FETCH_SIZE"), per(sys.argv[2], "WRITE_SIZE")
out = {}
for (grid, wg) in sorted(f):
    n = grid * 4  # radix-4 groups: one thread per four elements of the tile
    out[f"ntt_pass_kernel n=2^{n.bit_length() - 1}"] = {"grid": grid, "workgroup": wg, "fetch_kib_raw": f[(grid, wg)], "write_kib": w.get((grid, wg), 0.0),
        "traffic_bytes_per_pass": (2 * f[(grid, wg)] + w.get((grid, wg), 0.0)) * 1024, "algorithmic_bytes_per_transform": 64 * n}
open("gpurun_out/r3p/ntt_pmc.json", "w").write(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
PY
find $O -name "*counter_collection.csv" -size +2M -delete
ls -la $O | head -40; tail -c 1500 $O/bench_full.json; cat $O/shard_timeline.txt | tail -16; cat $O/msm_2_20_timeline.txt | tail -24
