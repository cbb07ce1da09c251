"""Worker of tests/test_gpu_dist.py::test_two_ranks_sharded_prove_equals_single_rank (launched by torchrun, gloo,
both ranks on cuda:0): the MSM-sharded prover session must produce, on every rank, exactly the proof a single
rank holding the whole SRS produces."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import typlonk_amd  # noqa: E402
from typlonk_amd.circuits import SquaringChain  # noqa: E402
from typlonk_amd.dist import ShardedMsm, ShardedProver  # noqa: E402
from bench import fr_mont_limbs, synthetic_scalars  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    log_n = int(os.environ.get("LOG_N", "10"))
    n = 1 << log_n
    ctx = typlonk_amd.Context(0)
    secret = fr_mont_limbs(0x5EC2E7)
    sh = ShardedMsm(ctx, n + 3, rank, world, torch.device("cpu"))
    sh.generate_srs(secret)
    if os.environ.get("TABLES") == "auto" and (sh.hi - sh.lo) >= (1 << 14):
        ctx.srs_precompute(sh.sid, 0)      # the library's choice by shard length (c = 15 below 2^16 points, 17 below 2^19)
    chain = SquaringChain(ctx, log_n)
    ch = [fr_mont_limbs(0xABC0 + k) for k in range(5)]
    args = (chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets, lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))
    sp = ShardedProver(sh)
    six = sp.prove(*args)
    bat = sp.prove(*args, challenge_v=lambda e: ch[4])
    # short MSMs: ranks whose index range is empty contribute the identity
    v = synthetic_scalars(n, 77, torch.device("cuda", 0))
    short = [sh.msm_devptr(v.data_ptr(), m) for m in (0, 1, 5, n // 2 + 1, n)]
    ok = True
    if rank == 0:
        full = ctx.srs_generate(secret, n + 3)
        ref6 = ctx.prove(full, *args)
        refb = ctx.prove(full, *args, challenge_v=lambda e: ch[4])
        same = lambda a, b: bool((np.asarray(a[0]) == np.asarray(b[0])).all() and int(a[1]) == int(b[1]))  # noqa: E731
        for got, ref in ((six, ref6), (bat, refb)):
            for key in ("commit", "t_commit", "witness"):
                ok &= len(got[key]) == len(ref[key]) and all(same(a, b) for a, b in zip(got[key], ref[key]))
            ok &= same(got["z_commit"], ref["z_commit"])
            ok &= all((a == b).all() for a, b in zip(got["evals"], ref["evals"]))
            ok &= bool((got["evals"][5] == 0).all())
        for m, got in zip((0, 1, 5, n // 2 + 1, n), short):
            ok &= same(got, ctx.msm_devptr(full, v.data_ptr(), m))
        print(json.dumps({"sharded_prove_ok": bool(ok), "world": world, "log_n": log_n}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 -- the launcher's summary buries a rank's traceback: one marked line for the test
        print(f"WORKER_ERROR rank={os.environ.get('RANK')}: {type(e).__name__}: {e}", flush=True)
        raise
