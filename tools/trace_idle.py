"""GPU busy / idle time from a rocprofv3 kernel trace (CSV): union of the kernel intervals between the first and the last
kernel of each proof-sized burst.  usage: python tools/trace_idle.py <kernel_trace.csv> [gap_ms_that_separates_bursts]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))))
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 3e6
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-50:]) > gap:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
for b in bursts:
    t0, t1 = b[0][0], max(x[1] for x in b)
    busy, end = 0, t0
    idle_gaps = []
    for s, e, _ in b:
        if s > end:
            idle_gaps.append(s - end)
            busy += e - s
            end = e
        elif e > end:
            busy += e - end
            end = e
    span = (t1 - t0) / 1e6
    if span < 1.0:
        continue
    big = sorted(idle_gaps, reverse=True)[:5]
    print(f"burst: {len(b):5d} kernels, span {span:8.3f} ms, busy {busy/1e6:8.3f} ms, idle {span - busy/1e6:7.3f} ms; largest gaps (us): {[round(g/1e3,1) for g in big]}")
