"""Timeline of the last proof in a rocprofv3 kernel trace (CSV): one line per kernel with its start offset, duration and
stream, so that what runs beside what -- and where the chip waits -- can be read off.
usage: python tools/trace_timeline.py <kernel_trace.csv> [from_ms to_ms]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ty::", ""), r["Stream_Id"])
               for r in csv.DictReader(open(sys.argv[1]))))
# bursts separated by > 3 ms of nothing
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-80:]) > 3e6:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
b = [x for x in bursts if (max(y[1] for y in x) - x[0][0]) > 5e6][-1]
t0 = b[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
streams = {s: i for i, s in enumerate(sorted({x[3] for x in b}))}
end = t0
for s, e, name, st in b:
    off = (s - t0) / 1e6
    gap = (s - end) / 1e3
    if lo <= off <= hi:
        print(f"{off:8.3f} ms  +{(e - s) / 1e3:8.1f} us  s{streams[st]}  {name[:40]:40s}" + (f"   <- idle {gap:.0f} us" if gap > 15 else ""))
    end = max(end, e)
print(f"span {(max(x[1] for x in b) - t0) / 1e6:.3f} ms, {len(b)} kernels, streams {len(streams)}")
