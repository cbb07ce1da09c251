"""Where the time of one rank of an 8-way sharded 2^20 MSM goes (run on one GPU as rank r of WORLD).

Prints wall time per local MSM, the kernel stage times, and the cost of the exchange path (all-gather + fold) measured
with a one-rank RCCL group (TYPLONK_FORCE_COLLECTIVE-style), which has the launch and copy latencies of the real thing
but no wire time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import typlonk_amd
from typlonk_amd.dist import ShardedMsm, allgather_fold, local_range
from bench import synthetic_scalars, fr_mont_limbs

log_n = int(os.environ.get("LOG_N", "20"))
world = int(os.environ.get("WORLD", "8"))
rank = int(os.environ.get("SHARD_RANK", "0"))
reps = int(os.environ.get("REPS", "100"))
n = 1 << log_n
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = typlonk_amd.Context(0)
ctx.set_profiling(True)
sh = ShardedMsm(ctx, n + 3, rank, world, dev)
sh.generate_srs(fr_mont_limbs(2))
# TABLES: window bits of the fixed-base tables; "auto" = the library's choice by shard length, "0" = none
tables = os.environ.get("TABLES", "auto")
if (sh.hi - sh.lo) >= (1 << 16) and tables != "0":
    ctx.srs_precompute(sh.sid, 0 if tables == "auto" else int(tables))
full = synthetic_scalars(n, 0x5EED0000 + log_n, dev)
for _ in range(10):
    sh.msm_local_devptr(full.data_ptr(), n)
stages = {}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    xy, inf = sh.msm_local_devptr(full.data_ptr(), n)
    for k, v in ctx.profile():
        stages[k] = stages.get(k, 0.0) + v
wall = (time.perf_counter() - t0) / reps * 1e3
out = {"log_n": log_n, "world": world, "rank": rank, "tables": tables, "lanes": os.environ.get("TYPLONK_MSM_LANES", "auto"),
       "reduce": os.environ.get("TYPLONK_MSM_REDUCE", "rc2"), "local_terms": local_range(n, n + 3, world, rank)[1] - local_range(n, n + 3, world, rank)[0],
       "local_msm_wall_ms": round(wall, 4), "stages_ms": {k: round(v / reps, 4) for k, v in stages.items()}}
ctx.set_profiling(False)
for _ in range(10):
    sh.msm_local_devptr(full.data_ptr(), n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    xy, inf = sh.msm_local_devptr(full.data_ptr(), n)
out["local_msm_wall_noprof_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)

if os.environ.get("NO_EXCHANGE") == "1":
    print("SHARD " + json.dumps(out))
    sys.exit(0)
# the exchange behind the C ABI on a one-rank communicator (RCCL all-gather of 104-B records + host fold): the launch,
# copy and synchronisation latencies of the real thing, no wire time
from typlonk_amd.capi import comm_unique_id
ctx.comm_init(comm_unique_id(), 0, 1)
for _ in range(10):
    ctx.comm_fold([(xy, inf)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.comm_fold([(xy, inf)])
out["native_fold_1_point_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
t0 = time.perf_counter()
for _ in range(reps):
    ctx.comm_fold([(xy, inf)] * 9)
out["native_fold_9_points_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
for _ in range(10):
    ctx.msm_sharded_devptr(sh.sid, full.data_ptr(), n)
t0 = time.perf_counter()
for _ in range(reps):
    ctx.msm_sharded_devptr(sh.sid, full.data_ptr(), n)
out["sharded_msm_wall_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
# the BATCHED shard rate: nine MSMs in flight per rank (a prover round's worth, three lanes) and ONE exchange for all nine --
# the mode in which the latency tails of a short MSM (sort, reduction, host finish, exchange) hide behind other MSMs'
# accumulations; prove() issues its commitments this way
nb = 9
ptrs, ms = [full.data_ptr()] * nb, [n - (k % 3) for k in range(nb)]
for _ in range(3):
    ctx.msm_sharded_batch_devptr(sh.sid, ptrs, ms)
t0 = time.perf_counter()
breps = max(3, reps // 10)
for _ in range(breps):
    ctx.msm_sharded_batch_devptr(sh.sid, ptrs, ms)
out["sharded_batch9_ms_per_msm"] = round((time.perf_counter() - t0) / breps / nb * 1e3, 4)
# what the one-rank communicator cannot show: the host fold over `world` ranks' records (typlonk_g1_fold_records_host is a
# host function: timed here on synthetic records, k * G per rank).  fold_extra_* = its cost above the one-rank fold the
# timings above already contain; the *_projected figures add it.
import numpy as np
from typlonk_amd.capi import g1_fold_records_host
def _fold_us(w, cnt):
    rec = np.zeros((w * cnt, 13), dtype=np.uint64)
    base = np.concatenate([np.asarray(xy, dtype=np.uint64).reshape(12), [np.uint64(inf)]])
    rec[:] = base                      # the same point on every rank and slot: the fold then doubles -- as costly as adding
    g1_fold_records_host(rec, w, cnt)
    ts = []
    for _ in range(200):
        t = time.perf_counter()
        g1_fold_records_host(rec, w, cnt)
        ts.append((time.perf_counter() - t) * 1e6)
    return sorted(ts)[len(ts) // 2]     # the median: a pre-empted call costs milliseconds
f1, f9, w1, w9 = _fold_us(1, 1), _fold_us(1, 9), _fold_us(world, 1), _fold_us(world, 9)
out["host_fold_us"] = {"one_rank_1_point": round(f1, 1), "one_rank_9_points": round(f9, 1), f"{world}_ranks_1_point": round(w1, 1),
                       f"{world}_ranks_9_points": round(w9, 1)}
out["sharded_msm_wall_projected_ms"] = round(out["sharded_msm_wall_ms"] + (w1 - f1) * 1e-3, 4)
out["sharded_batch9_ms_per_msm_projected"] = round(out["sharded_batch9_ms_per_msm"] + (w9 - f9) * 1e-3 / nb, 4)
# one rank's share of a sharded prove(): NTTs, grand product and quotient are replicated, the 13 commitments run on the
# shard, three exchanges (on the one-rank communicator) -- the per-rank time of BASELINE config 4 / 5 without wire time
if os.environ.get("NO_PROVE") != "1":
    from typlonk_amd.circuits import SquaringChain
    chain = SquaringChain(ctx, log_n)
    for _ in range(2):
        ctx.prove_native(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.prove_native(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
    out["prove_on_shard_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    chain.free()
print("SHARD " + json.dumps(out))
