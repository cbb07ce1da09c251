# PMC traffic of the NTT passes (2^20 and 2^22, forward plain): separate FETCH_SIZE / WRITE_SIZE passes
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
SIZES=20,22 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/ntt_bench.py > $O/f.log 2>&1
SIZES=20,22 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/ntt_bench.py > $O/w.log 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 - "$F" "$W" <<'PY'
import csv, sys, collections, json
def per(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("ty::ntt_pass_kernel"):
            agg[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f, w = per(sys.argv[1], "FETCH_SIZE"), per(sys.argv[2], "WRITE_SIZE")
out = {}
for grid in sorted(f):
    n = grid // 256 * 1024  # 256 threads per 1024-element tile
    out[f"ntt_pass_kernel n=2^{n.bit_length() - 1}"] = {"grid": grid, "fetch_kib_raw": f[grid], "write_kib": w.get(grid, 0.0),
        "traffic_bytes_per_pass": (2 * f[grid] + w.get(grid, 0.0)) * 1024, "algorithmic_bytes_per_transform": 64 * n}
print(json.dumps(out, indent=1))
open("gpurun_out/r2n/ntt_pmc.json", "w").write(json.dumps(out, indent=1) + "\n")
PY
