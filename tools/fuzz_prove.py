"""Time-boxed differential fuzz of the whole prover against the all-core CPU prover (oracle/cpu_prover.py).

Random domain sizes 2^3 .. 2^FUZZ_MAX_LOG, witnesses (start value, blinding rows), SRS secrets, challenges, table
choices; every proof element -- 7 commitments, 6 witnesses, 6 evaluations -- must be equal.  Not part of the suite."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, typlonk_amd
from oracle import cpu_prover as CP
from oracle import bls12_381 as O
from typlonk_amd.circuits import SquaringChain

SECONDS = float(os.environ.get("FUZZ_SECONDS", "600"))
SEED = int(os.environ.get("FUZZ_SEED", "1"))
MAX_LOG = int(os.environ.get("FUZZ_MAX_LOG", "15"))
rng = np.random.default_rng(SEED)
ctx = typlonk_amd.Context(0)


def limbs(v):
    return np.array(O.fr_to_mont_limbs(int(v) % O.R), dtype=np.uint64)


stats, t_end = {"proofs": 0, "host_column_proofs": 0, "fail": 0, "by_log_n": {}}, time.time() + SECONDS
while time.time() < t_end:
    log_n = int(rng.integers(3, MAX_LOG + 1))
    n = 1 << log_n
    sid = ctx.srs_generate(limbs(int(rng.integers(2, 1 << 62))), n + 3)
    tables = int(rng.choice([-1, 0, 0, 15, 17, 20]))
    if tables >= 0:
        try:
            ctx.srs_precompute(sid, tables)
        except typlonk_amd.TyplonkError:
            tables = -1
    srs_xy, srs_inf = ctx.srs_download(sid)
    chain = SquaringChain(ctx, log_n, x0=int(rng.integers(2, 1 << 62)), blinder_seed=int(rng.integers(0, 1 << 30)), keep_host=True)
    inputs = chain.host_inputs()
    for _ in range(2):
        ch = [limbs(int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62))) for _ in range(4)]
        proof = ctx.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,
                          lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))
        ref = CP.prove(log_n, inputs, srs_xy, srs_inf, ch)
        ok = all((np.asarray(a[0]) == np.asarray(b[0])).all() and int(a[1]) == int(b[1])
                 for k in ("commit", "t_commit", "witness") for a, b in zip(proof[k], ref[k]))
        ok = ok and (np.asarray(proof["z_commit"][0]) == np.asarray(ref["z_commit"][0])).all()
        ok = ok and all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(proof["evals"], ref["evals"]))
        stats["proofs"] += 1
        stats["by_log_n"][log_n] = stats["by_log_n"].get(log_n, 0) + 1
        if not ok:
            stats["fail"] += 1
            print(f"FAIL prove log_n={log_n} tables={tables} seed={SEED}", flush=True)
    # the native proof (transcript challenges) from device buffers against the one from host columns
    dev_proof = ctx.prove_native(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
    host_proof = ctx.prove_native_host(sid, chain.circuit, [b.download() for b in chain.wire_evals],
                                       chain.pi_evals.download() if chain.pi_evals is not None else None, chain.cosets)
    ok = all((np.asarray(a[0]) == np.asarray(b[0])).all() and int(a[1]) == int(b[1])
             for k in ("commit", "t_commit", "witness") for a, b in zip(host_proof[k], dev_proof[k]))
    ok = ok and (np.asarray(host_proof["z_commit"][0]) == np.asarray(dev_proof["z_commit"][0])).all()
    ok = ok and all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(host_proof["evals"], dev_proof["evals"]))
    stats["host_column_proofs"] += 1
    if not ok:
        stats["fail"] += 1
        print(f"FAIL prove_host log_n={log_n} tables={tables} seed={SEED}", flush=True)
    chain.free()
    ctx.srs_free(sid)
ctx.close()
stats["by_log_n"] = dict(sorted(stats["by_log_n"].items()))
print(f"prove fuzz seed {SEED}, {SECONDS:.0f} s: {stats}")
sys.exit(1 if stats["fail"] else 0)
