"""Kernel timeline of ONE steady-state MSM out of a rocprofv3 kernel trace (CSV) of tools/shard_latency.py or
tools/msm_loop.py: the kernels up to and including the device-to-host copy that ends an MSM, averaged over the last
`reps` MSMs: start offset, duration, gap to the previous kernel's end.
usage: python tools/msm_timeline.py <kernel_trace.csv> [reps]"""
import csv, sys
from collections import defaultdict

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ty::", "").split("<")[0])
              for r in csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
# an MSM ends with the kernel that writes its bit planes to the host, or with the device-to-host copy of its window sums
ends = [i for i, r in enumerate(rows)
        if r[2] in ("__amd_rocclr_copyBuffer", "msm_rc2_planes_kernel") and (i + 1 == len(rows) or rows[i + 1][2] != "__amd_rocclr_copyBuffer")]
starts = [e + 1 for e in ends][-(reps + 1):]
seqs = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
shape = tuple(r[2] for r in seqs[-1])
seqs = [s for s in seqs if tuple(r[2] for r in s) == shape]
n = len(seqs)
print(f"{n} MSMs of {len(shape)} kernels each")
tot_busy = 0.0
for k, name in enumerate(shape):
    off = sum(s[k][0] - s[0][0] for s in seqs) / n / 1e3
    dur = sum(s[k][1] - s[k][0] for s in seqs) / n / 1e3
    gap = sum(s[k][0] - max(x[1] for x in s[:k]) for s in seqs) / n / 1e3 if k else 0.0
    end_off = sum(s[k][1] - s[0][0] for s in seqs) / n / 1e3
    tot_busy += dur
    print(f"{off:8.1f} us  +{dur:7.1f} us  -> {end_off:8.1f}  gap {gap:6.1f}  {name}")
span = sum(max(x[1] for x in s) - s[0][0] for s in seqs) / n / 1e3
period = (rows[min(starts[-1], len(rows) - 1)][0] - rows[starts[0]][0]) / (len(starts) - 1) / 1e3 if len(starts) > 1 else 0
print(f"kernel span {span:.1f} us, kernel time {tot_busy:.1f} us, MSM period {period:.1f} us")
