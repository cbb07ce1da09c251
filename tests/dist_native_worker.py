"""Worker of tests/test_gpu_dist.py::test_native_exchange_with_world_N_on_one_gpu -- one FRESH process per rank, no torch,
no torch.distributed: the library's own exchange (typlonk_comm_*, comm.hip) with world > 1, all ranks on GPU 0, carried
by the test-only stand-in tests/cpp/libfake_rccl.so (TYPLONK_RCCL_LIB).
usage: dist_native_worker.py <rank> <world> <scratch dir> <log_n>
Writes <dir>/rank<r>.json; the parent compares every rank's points with a single-rank run, bit for bit."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import typlonk_amd  # noqa: E402
from typlonk_amd.capi import ERR_COMM, ERR_HIP, ERR_LENGTH, TyplonkError, comm_unique_id  # noqa: E402
from typlonk_amd.circuits import SquaringChain, fr_mont_limbs  # noqa: E402

SECRET = 0x5EC2E7


def scalars(n, seed):
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return sc


def batch_lengths(n):
    """40 MSMs (more than the 32 records of one exchange piece): full, n-1, n-3, empty, and short ones whose index range
    is empty on the higher ranks"""
    base = [n, n - 1, n - 3, 0, 1, 5, n // 2 + 1, n // 8]
    return [base[i % 8] if i < 32 else max(0, n - 7 * i) for i in range(40)]


def pt(p):
    return [[int(v) for v in np.asarray(p[0]).reshape(12)], int(p[1])]


def main():
    rank, world, d, log_n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    n = 1 << log_n
    total = n + 3
    ctx = typlonk_amd.Context(0)
    uid_path = os.path.join(d, "uid.bin")
    if rank == 0:
        uid = comm_unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    ctx.comm_init(uid, rank, world)
    assert ctx.comm_info() == (rank, world)
    out = {"rank": rank, "world": world}

    # ---- a staging failure on the LAST rank in the very first fold (TYPLONK_TEST_COMM_FAIL_STAGING=1 there) -------------
    g = (np.array([0] * 12, dtype=np.uint64), 1)
    try:
        ctx.comm_fold([g])
        out["staging"] = "ok"
    except TyplonkError as e:
        out["staging"] = [e.code, str(e)]
    out["staging_next"] = pt(ctx.comm_fold([g])[0])       # the communicator is still usable

    lo, hi = rank * total // world, (rank + 1) * total // world
    sid = ctx.srs_generate(fr_mont_limbs(SECRET), hi - lo, start=lo)
    ctx.srs_set_shard(sid, lo, total)
    if os.environ.get("TABLES") == "1" and hi - lo >= (1 << 14):
        ctx.srs_precompute(sid, 0)
    buf = ctx.alloc(n)
    buf.upload(scalars(n, 77))
    # ---- one MSM, the batch in pieces, the fold of caller-held points --------------------------------------------------------
    out["msm"] = [pt(ctx.msm_sharded_devptr(sid, buf.devptr, m)) for m in (n, n - 1, 1, 0)]
    ms = batch_lengths(n)
    out["batch"] = [pt(p) for p in ctx.msm_sharded_batch_devptr(sid, [buf.devptr] * len(ms), ms)]
    part = ctx.msm_devptr(sid, buf.devptr, n)            # this rank's partial sum, folded by hand
    out["fold"] = pt(ctx.comm_fold([part])[0])
    # ---- the failure path: rank 1 asks for more terms than the SRS has ---------------------------------------------------------
    try:
        ctx.msm_sharded_devptr(sid, buf.devptr, total + 1 if rank == 1 else n)
        out["fail"] = "ok"
    except TyplonkError as e:
        out["fail"] = [e.code, str(e)]
    out["fail_next"] = pt(ctx.msm_sharded_devptr(sid, buf.devptr, n))
    # ... and in a batch: the whole group fails together
    try:
        ctx.msm_sharded_batch_devptr(sid, [buf.devptr] * 3, [n, total + 1 if rank == 1 else n, 5])
        out["fail_batch"] = "ok"
    except TyplonkError as e:
        out["fail_batch"] = [e.code, str(e)]
    # ---- typlonk_prove on the shard: three collectives per proof, all ranks the same proof ------------------------------------
    chain = SquaringChain(ctx, log_n)
    pr = ctx.prove_native(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
    out["proof"] = {"commit": [pt(p) for p in pr["commit"]], "z": pt(pr["z_commit"]), "t": [pt(p) for p in pr["t_commit"]],
                    "w": [pt(p) for p in pr["witness"]], "evals": [[int(v) for v in e] for e in pr["evals"]],
                    "ch": {k: [int(v) for v in val] for k, val in pr["challenges"].items()}}
    out["codes"] = {"comm": ERR_COMM, "hip": ERR_HIP, "length": ERR_LENGTH}
    ctx.comm_destroy()
    ctx.close()
    with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
