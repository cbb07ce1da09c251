"""world_size-2 gloo test of the multi-GPU MSM plumbing on CPU: index sharding, the all-gather of
partial points and the fixed-order fold (typlonk_g1_sum_host).  The local MSM is the CPU oracle
here (test-only stand-in for the HIP kernel, which needs a GPU); the exchange + fold is product code."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, total_len, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from helpers import O, fr_pack, g1_pack, g1_unpack_one
    from oracle import coracle as CO
    from typlonk_amd import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    srs = O.srs_from_secret_fast(3, total_len)
    xy, inf = g1_pack(srs)
    scalars = O.random_frs(0xD157, m)
    lo, hi = D.local_range(m, total_len, world, rank)
    sl, sh = D.shard_bounds(total_len, world, rank)
    assert hi == lo or (lo == sl and hi <= sh)   # a non-empty local range starts at the shard start
    pxy, pinf = CO.msm_reference(fr_pack(scalars[lo:hi]) if hi > lo else np.zeros((0, 4), dtype=np.uint64),
                                 xy[sl:sh], inf[sl:sh])
    out, oinf = D.allgather_fold(pxy, pinf, torch.device("cpu"))
    got = g1_unpack_one(out, oinf)
    exp = O.g1_mul(O.G1, O.poly_eval(scalars, 3))
    q.put((rank, got == exp, [int(x) for x in out], oinf))
    dist.barrier()
    dist.destroy_process_group()


def _run(m, total_len, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, total_len, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    # bit-identical on every rank
    assert len({(tuple(o), i) for _, _, o, i in res}) == 1


def test_sharded_msm_world2_full_length():
    _run(m=24, total_len=24)


def test_sharded_msm_world2_short_and_ragged():
    # m falls entirely inside rank 0's shard -> rank 1 contributes the identity
    _run(m=5, total_len=27)
    # ragged: 27 bases -> chunks of 14 / 13, m = 26 cuts rank 1's shard short
    _run(m=26, total_len=27)


def test_shard_bounds_cover_and_do_not_overlap():
    sys.path.insert(0, ROOT)
    from typlonk_amd import dist as D

    for total in (1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 4, 8):
            edges = [D.shard_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            for m in (0, 1, total // 2, total):
                assert sum(hi - lo for lo, hi in (D.local_range(m, total, world, r) for r in range(world))) == m
