"""The sharded (N > 1) MSM path of bench.py with real kernels: two ranks share the one GPU of the test box
and exchange their partial points over gloo (RCCL refuses two ranks per device; the RCCL path itself is
exercised with one rank by TYPLONK_FORCE_COLLECTIVE).  Checks the sharded result against the reference's
identity commit(p) == [p(s)]G and the JSON contract fields."""
import json
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(env_extra, nproc, args):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def test_two_ranks_sharded_msm_on_one_gpu():
    d = _run({"TYPLONK_BENCH_BACKEND": "gloo"}, 2, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--log-n", "18"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["metric"] == "msm_g1_adds_per_s"
    assert d["parity"]["full_commit_identity"] is True and d["value"] is not None
    assert "index-sharded x2" in d["config"]["parallelism"]
    assert d["msm_batch"]["msms"] == 9 and d["msm_batch"]["first_equals_single"] is True


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the driver types it (no torch.distributed.run around it): the script starts the two
    rank processes itself, and the contract line -- n_gpus = 2, the exchange named -- is the LAST line of its stdout."""
    env = dict(os.environ, TYPLONK_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "20", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out[-1].startswith('{"metric"') and sum(l.startswith('{"metric"') for l in out) == 1
    d = json.loads(out[-1])
    assert d["n_gpus"] == 2 and d["value"] is not None and "error" not in d
    assert d["parity"]["full_commit_identity"] is True
    assert d["rccl_world"] == 0 and "gloo" in d["exchange"] and "ranks_share_gpu" in d
    assert d["msm_batch"]["ms_per_msm"] > 0 and d["prove_sharded_ms"] > 0 and d["prove_sharded_batched_ms"] > 0
    # the three modes side by side, each against this round's one-GPU record, plus the one-GPU projection for this N
    # (bench.py, add_scaling_context): a SCALE record is readable without re-deriving DESIGN.md section 7
    sc = d["vs_committed_reference"]      # context only (another box's record); the ratios of THIS run are scaling_same_run
    assert sc["standalone"] > 0 and sc["batched"] > 0 and sc["prove"] > 0 and "CONTEXT ONLY" in sc["read_as"]
    assert "batched" in sc["claim"] and d["one_gpu_reference"]["source"].startswith("profiles/r0")
    same = d["scaling_same_run"]
    assert same["standalone"] > 0 and same["batched"] > 0 and d["one_gpu_same_run"]["ms_per_step"] > 0
    exp = d["expected_from_1gpu"]
    assert exp["speedup"]["batched_msms"] > exp["speedup"]["one_msm"] > 1 and "shard_latency" in exp["source"]


def test_one_rank_through_rccl_all_gather():
    d = _run({"TYPLONK_FORCE_COLLECTIVE": "1"}, 1, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "16",
                                                    "--cpu-sample", "2048"])
    assert d["n_gpus"] == 1 and d["parity"]["sample_vs_oracle"] and d["parity"]["full_commit_identity"]
    assert d["prove_valid"] is True and d["rccl_world"] == 1 and "rccl" in d["exchange"]
    for key in ("roofline", "cpu_baseline", "ms_per_step", "higher_is_better", "dtype", "data", "config"):
        assert key in d


def test_two_ranks_sharded_prove_equals_single_rank():
    """typlonk_srs_set_shard + ShardedProver: both proof shapes, every element, plus short MSMs whose range is
    empty on one rank (tests/dist_prove_worker.py)"""
    env = dict(os.environ, LOG_N="10")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_prove_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"sharded_prove_ok"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(lines[0])["sharded_prove_ok"] is True


def test_config4_eight_ranks_share_the_gpu_at_2_20():
    """BASELINE config 4 at its own size on the one GPU of the test box: the 2^20-row circuit with every MSM index-sharded
    over EIGHT ranks (gloo: RCCL refuses several ranks per device; each rank has its own context, SRS shard of 2^17
    points and c = 17 tables).  bench.py's headline loop: the folded commitment equals [p(s)]G; the worker: the sharded
    proof -- both shapes -- equals the proof of one rank holding the whole SRS, element for element, r(zeta) = 0."""
    d = _run({"TYPLONK_BENCH_BACKEND": "gloo", "TYPLONK_BENCH_PG_TIMEOUT": "600"}, 8,
             ["--gpus", "8", "--steps", "3", "--warmup", "1", "--log-n", "20"])
    assert d["n_gpus"] == 8 and d["parity"]["full_commit_identity"] is True and d["value"] is not None
    assert d["prove_valid"] is True and "prove_sharded_error" not in d and "error" not in d
    assert "c=17" in d["config"]["fixed_base_tables"]
    env = dict(os.environ, LOG_N="20", TABLES="auto")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_prove_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"sharded_prove_ok"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["sharded_prove_ok"] is True and out["world"] == 8 and out["log_n"] == 20


def test_native_rccl_exchange_behind_the_c_abi(built):
    """typlonk_comm_* (RCCL loaded by the library itself, no torch.distributed anywhere): a one-rank communicator on the
    test box's GPU.  typlonk_msm_g1_sharded_devptr / _batch_devptr and typlonk_comm_fold_g1 return what the local calls
    return (the fold over one rank is the identity map on points, through ncclAllGather and the host fold), and
    typlonk_prove on an SRS shard -- which folds every round's commitments -- equals typlonk_prove on the plain SRS."""
    import numpy as np

    import typlonk_amd
    from helpers import O
    from typlonk_amd.capi import TyplonkError, ERR_INVALID_ARG, comm_unique_id
    from typlonk_amd.circuits import SquaringChain

    ctx = typlonk_amd.Context(0)
    try:
        assert ctx.comm_info()[1] == 0
        with pytest.raises(TyplonkError) as e:          # no communicator yet
            ctx.comm_fold([(np.zeros(12, dtype=np.uint64), 1)])
        assert e.value.code == ERR_INVALID_ARG
        uid = comm_unique_id()
        assert len(uid) == 128
        ctx.comm_init(uid, 0, 1)
        assert ctx.comm_info() == (0, 1)
        log_n = 12
        n = 1 << log_n
        s_limbs = np.array(O.fr_to_mont_limbs(0xC0FFEE), dtype=np.uint64)
        plain = ctx.srs_generate(s_limbs, n + 3)
        shard = ctx.srs_generate(s_limbs, n + 3)
        ctx.srs_set_shard(shard, 0, n + 3)
        part = ctx.srs_generate(s_limbs, 1000, start=500)          # a proper sub-range
        ctx.srs_set_shard(part, 500, n + 3)
        rng = np.random.default_rng(5)
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        buf = ctx.alloc(n)
        buf.upload(sc)
        a = ctx.msm_devptr(plain, buf.devptr, n)
        b = ctx.msm_sharded_devptr(shard, buf.devptr, n)
        assert (a[0] == b[0]).all() and a[1] == b[1]
        loc = ctx.msm_devptr(part, buf.devptr, n)
        fol = ctx.msm_sharded_devptr(part, buf.devptr, n)
        assert (loc[0] == fol[0]).all() and loc[1] == fol[1]
        res = ctx.msm_sharded_batch_devptr(shard, [buf.devptr] * 3, [n, n - 1, 0])
        ref = ctx.msm_batch_devptr(plain, [buf.devptr] * 3, [n, n - 1, 0])
        for (x, i), (y, j) in zip(res, ref):
            assert (x == y).all() and i == j
        assert res[2][1] == 1                                         # the empty sum is the identity, also after the fold
        # a rank whose local MSM fails still joins the collective (flagged records) and reports ITS error; the
        # communicator stays usable
        from typlonk_amd.capi import ERR_LENGTH
        with pytest.raises(TyplonkError) as e:
            ctx.msm_sharded_devptr(shard, buf.devptr, n + 100)        # m > total length: kzg/src/lib.rs:43
        assert e.value.code == ERR_LENGTH
        again = ctx.msm_sharded_devptr(shard, buf.devptr, n)
        assert (again[0] == b[0]).all() and again[1] == b[1]
        pts = [a, loc, (np.zeros(12, dtype=np.uint64), 1)] * 24       # 72 points: three pieces of the 32-record exchange buffers
        back = ctx.comm_fold(pts)
        for (x, i), (y, j) in zip(pts, back):
            assert i == j and (j == 1 or (x == y).all())
        chain = SquaringChain(ctx, log_n)
        p0 = ctx.prove_native(plain, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        p1 = ctx.prove_native(shard, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        for key in ("commit", "t_commit", "witness"):
            for (x, i), (y, j) in zip(p0[key], p1[key]):
                assert (x == y).all() and i == j, key
        assert (p0["z_commit"][0] == p1["z_commit"][0]).all()
        for x, y in zip(p0["evals"], p1["evals"]):
            assert (x == y).all()
        chain.free()
        buf.free()
        ctx.comm_destroy()
        assert ctx.comm_info()[1] == 0
    finally:
        ctx.close()


@pytest.mark.slow
def test_config5_eight_ranks_prove_at_2_22_batched_and_six_openings():
    """BASELINE config 5 as written, on the one GPU of the test box: the 2^22-row circuit, full prove() with the quotient's
    2^24-point coset NTTs, every MSM index-sharded over EIGHT ranks (gloo; 2^19-point SRS shards with the library's own
    table choice), in the reference's proof shape AND with batched KZG openings -- each equal, element for element, to
    the proof of one rank holding the whole SRS, r(zeta) = 0; plus short MSMs whose range is empty on most ranks."""
    from conftest import need_resources

    # eight ranks on ONE GPU, each with a 2^19-point shard + tables, the 2^22-row circuit's coset evaluations (4.8 GiB), a
    # 2^24-point quotient workspace and the NTT tables of both sizes: ~17 GiB per rank, 134 GB measured in all (round 6)
    need_resources(host_gib=24, hbm_gib=150)
    env = dict(os.environ, LOG_N="22", TABLES="auto")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_prove_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"sharded_prove_ok"')]
    errs = [l for l in (r.stdout + r.stderr).splitlines() if "WORKER_ERROR" in l]
    assert r.returncode == 0 and len(lines) == 1, "\n".join(errs) + r.stdout[-1000:] + r.stderr[-1000:]
    out = json.loads(lines[0])
    assert out["sharded_prove_ok"] is True and out["world"] == 8 and out["log_n"] == 22


def _native_ranks(world, log_n, tmp_path, tables=False, timeout=900):
    """start `world` fresh rank processes of tests/dist_native_worker.py on GPU 0 over the stand-in exchange library"""
    fake = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
    assert os.path.exists(fake), "tests/cpp/libfake_rccl.so not built (__graft_entry__.build())"
    procs = []
    for r in range(world):
        env = dict(os.environ, TYPLONK_RCCL_LIB=fake, FAKE_RCCL_TIMEOUT_S="300", TABLES="1" if tables else "0")
        env.pop("TYPLONK_TEST_COMM_FAIL_STAGING", None)
        if r == world - 1:
            # this rank's FIRST fold loses its staging copy: the fault injection exists only in the test build of the
            # library (-DTYPLONK_TEST_HOOKS); every other rank runs the shipped one
            hooked = os.path.join(ROOT, "tests", "cpp", "hooks", "libtyplonk_hip.so")
            assert os.path.exists(hooked), "tests/cpp/hooks/libtyplonk_hip.so not built (__graft_entry__.build())"
            env["TYPLONK_LIB_PATH"] = hooked
            env["TYPLONK_TEST_COMM_FAIL_STAGING"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_native_worker.py"), str(r), str(world),
                                       str(tmp_path), str(log_n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True, cwd=ROOT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    return [json.load(open(os.path.join(str(tmp_path), f"rank{r}.json"))) for r in range(world)]


@pytest.mark.parametrize("world,log_n,tables", [(2, 10, False), (8, 12, False), (2, 15, True)])
def test_native_exchange_with_world_N_on_one_gpu(built, tmp_path, world, log_n, tables):
    """The library's OWN exchange with more than one rank (round 4 had only ever run it with world = 1): `world` fresh,
    torch-free processes share GPU 0, each with its own context, SRS shard and communicator; ncclAllGather is carried by
    the test-only tests/cpp/libfake_rccl.so (RCCL refuses two ranks per device), selected through TYPLONK_RCCL_LIB -- the
    record staging, the rank-order fold, the piecewise path (40 points > 32 per exchange), typlonk_prove's three
    collectives per proof and both failure paths are the product's code, comm.hip.
      * every rank returns the single-rank result, bit for bit: MSMs of n, n-1, 1, 0 terms, a batch of 40, a fold of
        caller-held partial sums, the whole proof (commitments, evaluations, squeezed challenges);
      * rank 1 passes m > len: IT gets TYPLONK_ERR_LENGTH, every other rank TYPLONK_ERR_COMM naming rank 1, and the next
        call works (alone and inside a batch);
      * the last rank's first staging copy fails (TYPLONK_TEST_COMM_FAIL_STAGING): it still joins the collective -- its
        poisoned send buffer goes out -- so it gets TYPLONK_ERR_HIP, the others TYPLONK_ERR_COMM, nobody hangs."""
    import numpy as np

    import typlonk_amd
    from dist_native_worker import SECRET, batch_lengths, pt, scalars
    from typlonk_amd.capi import ERR_COMM, ERR_HIP, ERR_LENGTH
    from typlonk_amd.circuits import SquaringChain, fr_mont_limbs

    ranks = _native_ranks(world, log_n, tmp_path, tables)
    n = 1 << log_n
    ctx = typlonk_amd.Context(0)
    try:
        full = ctx.srs_generate(fr_mont_limbs(SECRET), n + 3)
        buf = ctx.alloc(n)
        buf.upload(scalars(n, 77))
        want_msm = [pt(ctx.msm_devptr(full, buf.devptr, m)) for m in (n, n - 1, 1, 0)]
        ms = batch_lengths(n)
        want_batch = [pt(p) for p in ctx.msm_batch_devptr(full, [buf.devptr] * len(ms), ms)]
        chain = SquaringChain(ctx, log_n)
        pr = ctx.prove_native(full, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        ident = pt((np.zeros(12, dtype=np.uint64), 1))
        ident[0][6:] = [int(v) for v in ctx.msm_devptr(full, buf.devptr, 0)[0][6:]]     # the (0, 1, inf) encoding
        for r, o in enumerate(ranks):
            assert o["msm"] == want_msm and o["batch"] == want_batch and o["fold"] == want_msm[0], r
            assert o["fail_next"] == want_msm[0], r
            assert o["proof"]["commit"] == [pt(p) for p in pr["commit"]] and o["proof"]["z"] == pt(pr["z_commit"]), r
            assert o["proof"]["t"] == [pt(p) for p in pr["t_commit"]] and o["proof"]["w"] == [pt(p) for p in pr["witness"]], r
            assert o["proof"]["evals"] == [[int(v) for v in e] for e in pr["evals"]], r
            assert o["proof"]["ch"] == {k: [int(v) for v in val] for k, val in pr["challenges"].items()}, r
            # failure of rank 1
            for key in ("fail", "fail_batch"):
                code, msg = o[key]
                if r == 1:
                    assert code == ERR_LENGTH, (r, key, o[key])
                else:
                    assert code == ERR_COMM and "rank 1" in msg, (r, key, o[key])
            # staging failure of the last rank
            code, msg = o["staging"]
            if r == world - 1:
                assert code == ERR_HIP and "staging" in msg, o["staging"]
            else:
                assert code == ERR_COMM and f"rank {world - 1}" in msg and "stage" in msg, o["staging"]
            assert o["staging_next"][1] == 1, r          # the sum of `world` identities, through a working exchange
    finally:
        ctx.close()


@pytest.mark.parametrize("world", [2, 8])
def test_native_exchange_world_N_from_plain_cpp_processes(built, tmp_path, world):
    """the same with no Python in the ranks: tests/cpp/test_comm_ranks_host forks `world` fresh copies of itself before
    touching the GPU; each is a rank on GPU 0 that checks the sharded MSMs (single, batch of 40) against a plain SRS it
    also holds, and the failure protocol (m > len on rank 1, a null output on rank 0)"""
    fake = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
    exe = os.path.join(ROOT, "tests", "cpp", "test_comm_ranks_host")
    env = dict(os.environ, TYPLONK_RCCL_LIB=fake, FAKE_RCCL_TIMEOUT_S="300")
    r = subprocess.run([exe, str(world), str(tmp_path)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and f"all {world} ranks ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
