#!/usr/bin/env python3
"""Arrival day of a multi-GPU box: read the contract lines of `python bench.py --gpus N` for N = 1, 2, 4, 8 and print, for
the three modes (one stand-alone MSM per step / nine MSMs in flight / the sharded prove()), what was MEASURED next to what
DESIGN.md section 7 projected from one GPU (`expected_from_1gpu`), plus the size of the RCCL communicator the library held
(`rccl_world`: N means the exchange ran inside the library over real RCCL).

usage:  python tools/scale_table.py line1.json line2.json ...     (files holding bench.py's last stdout line, any order)
        python tools/scale_table.py --run                         (runs bench.py --gpus 1, 2, 4, 8 itself, one after the other;
                                                                   each is a child process -- nothing here touches the GPU)
Speed-ups are this run's N-GPU time into this run's 1-GPU time when a 1-GPU line is given, else into the line's own
`one_gpu_same_run` (rank 0's unsharded MSM measured in the same process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(text):
    for line in reversed([l for l in text.splitlines() if l.strip()]):
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError("no JSON line")


def load(paths):
    lines = {}
    for p in paths:
        d = last_json_line(open(p).read())
        lines[int(d["n_gpus"])] = d
    return lines


def run_all():
    lines = {}
    for n in (1, 2, 4, 8):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, cwd=ROOT)
        if r.returncode != 0:
            print(f"bench.py --gpus {n} failed (rc {r.returncode}): {r.stderr[-400:]}", file=sys.stderr)
            continue
        lines[n] = last_json_line(r.stdout)
    return lines


def fmt(x, nd=3):
    return "-" if x is None else f"{x:.{nd}f}"


def main():
    lines = run_all() if sys.argv[1:] == ["--run"] else load(sys.argv[1:])
    if not lines:
        sys.exit("no bench lines")
    one = lines.get(1)
    base = {"standalone": one and one.get("ms_per_step"), "batched": one and (one.get("msm_batch") or {}).get("ms_per_msm"),
            "prove": one and (one.get("prove_native_ms") or one.get("prove_ms"))}
    print(f"{'N':>2} {'rccl_world':>10} | {'MSM ms':>8} {'x':>6} {'proj ms':>8} {'proj x':>6} | {'batch ms/MSM':>12} {'x':>6} {'proj ms':>8} {'proj x':>6} |"
          f" {'prove ms':>9} {'x':>6} {'proj ms':>8} {'proj x':>6}")
    for n in sorted(lines):
        d = lines[n]
        same = d.get("one_gpu_same_run") or {}
        b1 = {"standalone": base["standalone"] or same.get("ms_per_step"), "batched": base["batched"] or same.get("msm_batch_ms_per_msm"),
              "prove": base["prove"]}
        meas = {"standalone": d.get("ms_per_step"), "batched": (d.get("msm_batch") or {}).get("ms_per_msm"),
                "prove": d.get("prove_sharded_native_ms") or d.get("prove_sharded_ms") or d.get("prove_native_ms") or d.get("prove_ms")}
        exp = d.get("expected_from_1gpu") or {}
        proj = {"standalone": exp.get("one_msm_plus_exchange_ms"), "batched": exp.get("batched_ms_per_msm"), "prove": exp.get("prove_on_shard_ms")}
        psp = exp.get("speedup") or {}
        pmap = {"standalone": psp.get("one_msm"), "batched": psp.get("batched_msms"), "prove": psp.get("prove")}
        cells = []
        for mode in ("standalone", "batched", "prove"):
            sp = (b1[mode] / meas[mode]) if (b1[mode] and meas[mode]) else None
            cells.append(f"{fmt(meas[mode]):>{12 if mode == 'batched' else (9 if mode == 'prove' else 8)}} {fmt(sp, 2):>6} {fmt(proj[mode]):>8} {fmt(pmap[mode], 2):>6}")
        print(f"{n:>2} {d.get('rccl_world', 0):>10} | " + " | ".join(cells))
    print("\nx = measured speed-up over this run's one-GPU figure; proj = DESIGN.md section 7's projection from one GPU "
          "(tools/shard_latency.py).  The north-star's >= 6x at 8 GPUs is claimed for the batched mode.")
    bad = [n for n, d in lines.items() if n > 1 and d.get("rccl_world", 0) != n]
    if bad:
        print(f"NOTE: N = {bad}: rccl_world != N -- the exchange did not run over the library's own RCCL communicator "
              f"(see `exchange` in the line)")


if __name__ == "__main__":
    main()
