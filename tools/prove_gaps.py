"""Where a proof's time goes, from a rocprofv3 kernel trace (CSV) of tools/prove_loop.py run with GAP_MS >= 5:
the proofs are the bursts of kernels separated by > 4 ms of nothing; one complete proof (a burst with the modal kernel
count) is printed as a timeline, followed by the union of the intervals in which a bucket accumulation is in flight, the
accumulation-free stretches, and per-kernel totals.
usage: python tools/prove_gaps.py <kernel_trace.csv> [--brief]"""
import collections, csv, sys

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ty::", ""), r["Stream_Id"])
              for r in csv.DictReader(open(sys.argv[1])))
bursts, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - end > 4e6:
        bursts.append(cur)
        cur = [r]
    else:
        cur.append(r)
    end = max(end, r[1])
bursts.append(cur)
big = [b for b in bursts if len(b) > 60]
mode = collections.Counter(len(b) for b in big).most_common(1)[0][0]
b = [x for x in big if len(x) == mode][-1]
t0 = b[0][0]
span = (max(x[1] for x in b) - t0) / 1e6
streams = {s: i for i, s in enumerate(sorted({x[3] for x in b}))}
if "--brief" not in sys.argv:
    for s, e, name, st in b:
        print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e3:8.1f} us  s{streams[st]}  {name[:44]}")
acc = sorted((s, e) for s, e, n, _ in b if n.startswith("msm_accum"))
merged = []
for s, e in acc:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
busy = sum(e - s for s, e in merged) / 1e6
print(f"\nproof: {len(b)} kernels on {len(streams)} streams, span {span:.3f} ms")
print(f"accumulation in flight: {busy:.3f} ms in {len(acc)} launches (sum of their own durations {sum(e - s for s, e in acc) / 1e6:.3f} ms)")
print(f"accumulation-free: {span - busy:.3f} ms:")
prev = t0
for s, e in merged + [[max(x[1] for x in b), 0]]:
    if s - prev > 0.05e6:
        names = collections.Counter(n for s2, e2, n, _ in b if s2 < s and e2 > prev and not n.startswith("msm_accum"))
        print(f"  {(prev - t0) / 1e6:7.3f} .. {(s - t0) / 1e6:7.3f} ms  ({(s - prev) / 1e6:.3f})  " + ", ".join(f"{n} x{c}" for n, c in names.most_common(6)))
    prev = max(prev, e) if e else prev
tot = collections.Counter()
cnt = collections.Counter()
for s, e, n, _ in b:
    tot[n] += e - s
    cnt[n] += 1
print("per kernel (sum of durations, ms):")
for n, v in tot.most_common(14):
    print(f"  {n:36s} {cnt[n]:4d} x  {v / 1e6:8.3f}")
