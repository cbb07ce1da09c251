"""Multi-GPU MSM: one process per GPU, index-sharded bases and scalars, one tiny exchange.

    sum_{i<m} k_i P_i  =  sum_g ( sum_{i in shard g} k_i P_i )

Each rank keeps its shard of the SRS resident in HBM (fixed chunking of the base vector, so the
shard boundaries do not depend on the MSM length m), runs the Pippenger MSM locally and produces
one canonical affine partial point.  RCCL has no elliptic-curve reduction, so the "all-reduce" of
the north-star is an all-gather of the G 13-word records (12 limbs + infinity flag) followed by a
fold in fixed rank order 0..G-1 on every rank -- bit-identical everywhere.
Message size is 104 B per rank: pure latency, xGMI bandwidth is irrelevant.

The exchange itself lives BEHIND the C ABI (typlonk_comm_init, typlonk_msm_g1_sharded_devptr,
typlonk_comm_fold_g1, and typlonk_prove on a shard: include/typlonk.h): this module is a thin caller
that only carries the RCCL rendezvous id from rank 0 to the others over the torch.distributed group
the launcher already set up.  The torch.distributed data path below (`allgather_fold*`) remains for
the `gloo` backend: the world_size-2 CPU tests, and several ranks sharing ONE GPU on a 1-GPU box
(RCCL refuses two ranks on the same device).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .capi import Context, g1_sum_host


def shard_bounds(total_len: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of the base vector owned by `rank`: equal chunks of ceil(total_len / world)."""
    chunk = (total_len + world - 1) // world
    lo = min(rank * chunk, total_len)
    hi = min(lo + chunk, total_len)
    return lo, hi


def local_range(m: int, total_len: int, world: int, rank: int) -> tuple[int, int]:
    """portion [lo, hi) of an m-term MSM (m <= total_len) that falls into `rank`'s shard"""
    lo, hi = shard_bounds(total_len, world, rank)
    return min(lo, m), min(hi, m)


def allgather_fold(partial_xy: np.ndarray, partial_inf: int, device: torch.device, group=None):
    """all-gather one affine point per rank and fold them in rank order; returns (xy[12], inf)"""
    world = dist.get_world_size(group)
    rec = np.zeros(13, dtype=np.uint64)
    rec[:12] = np.asarray(partial_xy, dtype=np.uint64).reshape(12)
    rec[12] = partial_inf
    mine = torch.from_numpy(rec.view(np.int64).copy()).to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    allrec = torch.stack(out).cpu().numpy().view(np.uint64)
    return g1_sum_host(allrec[:, :12].copy(), allrec[:, 12].astype(np.uint8))


def allgather_fold_many(points, device: torch.device, group=None):
    """the same for k points at once (one collective per prover round): points = [(xy[12], inf), ...]"""
    world = dist.get_world_size(group)
    k = len(points)
    rec = np.zeros((k, 13), dtype=np.uint64)
    for i, (xy, inf) in enumerate(points):
        rec[i, :12] = np.asarray(xy, dtype=np.uint64).reshape(12)
        rec[i, 12] = inf
    mine = torch.from_numpy(rec.view(np.int64).copy()).to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    allrec = torch.stack(out).cpu().numpy().view(np.uint64)      # (world, k, 13)
    return [g1_sum_host(allrec[:, i, :12].copy(), allrec[:, i, 12].astype(np.uint8)) for i in range(k)]


class ShardedMsm:
    """SRS shard resident on this rank's GPU + the collective combine."""

    def __init__(self, ctx: Context, total_len: int, rank: int, world: int, device: torch.device, group=None):
        self.ctx, self.total_len, self.rank, self.world = ctx, total_len, rank, world
        self.device, self.group = device, group
        self.lo, self.hi = shard_bounds(total_len, world, rank)
        self.sid = None
        # exercise the all-gather + fold even with one rank (used to validate the RCCL path on a 1-GPU box)
        import os
        self.force_collective = os.environ.get("TYPLONK_FORCE_COLLECTIVE") == "1" and dist.is_initialized()
        # the native exchange (RCCL inside the library) whenever the process group runs on RCCL
        self.native = False
        if dist.is_initialized() and dist.get_backend(group) == "nccl" and (world > 1 or self.force_collective) \
                and os.environ.get("TYPLONK_NATIVE_COMM", "1") != "0":
            from .capi import comm_available, comm_unique_id
            # A rank on which the native communicator cannot come up must not leave the others inside a collective of
            # a communicator it is not part of: every step is agreed on by all ranks (MIN over a flag), and if any rank
            # failed, all of them fall back to the torch.distributed exchange below.
            def agree(ok: bool) -> bool:
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                return bool(t.item())

            box, err = [None], None
            # comm_init is itself a collective (ncclCommInitRank): every rank first confirms -- without entering one --
            # that it can load librccl at all (typlonk_comm_available), so that no rank waits inside comm_init alone
            if not comm_available():
                err = RuntimeError("librccl cannot be loaded on this rank")
            try:
                if rank == 0 and err is None:
                    box[0] = comm_unique_id()
            except Exception as e:  # noqa: BLE001 -- reported, not fatal
                err = e
            if agree(err is None):
                dist.broadcast_object_list(box, src=0, group=group, device=device)
                try:
                    ctx.comm_init(box[0], rank, world)
                except Exception as e:  # noqa: BLE001
                    err = e
                if agree(err is None):
                    self.native = True
                else:
                    try:
                        ctx.comm_destroy()
                    except Exception:  # noqa: BLE001
                        pass
            if not self.native and rank == 0:
                print(f"typlonk_amd.dist: native RCCL exchange unavailable ({err}); using torch.distributed", flush=True)

    def generate_srs(self, secret_limbs):
        """build only this rank's slice [s^lo G, ..., s^(hi-1) G] in HBM"""
        self.sid = self.ctx.srs_generate(secret_limbs, self.hi - self.lo, start=self.lo)
        self.ctx.srs_set_shard(self.sid, self.lo, self.total_len)
        return self.sid

    def load_srs(self, xy_full, inf_full=None):
        xy = np.asarray(xy_full, dtype=np.uint64).reshape(-1, 12)[self.lo:self.hi]
        inf = None if inf_full is None else np.asarray(inf_full, dtype=np.uint8)[self.lo:self.hi]
        self.sid = self.ctx.srs_load(xy, inf)
        self.ctx.srs_set_shard(self.sid, self.lo, self.total_len)
        return self.sid

    def msm_local_devptr(self, d_scalars: int, m: int):
        """this rank's partial sum of an m-term MSM; d_scalars = device address of coefficient 0 of the full
        vector (a rank holding only its slice passes slice_address - 32 * lo; nothing outside the slice is read)"""
        return self.ctx.msm_devptr(self.sid, d_scalars, m)

    def msm_devptr(self, d_scalars: int, m: int):
        """full m-term MSM result on every rank"""
        if self.native:
            return self.ctx.msm_sharded_devptr(self.sid, d_scalars, m)
        xy, inf = self.msm_local_devptr(d_scalars, m)
        if self.world == 1 and not self.force_collective:
            return xy, inf
        return allgather_fold(xy, inf, self.device, self.group)

    def msm_batch_devptr(self, d_scalars_list, ms):
        """several full MSMs over the same SRS (a prover round's commitments): the local partial sums run as one
        pipelined batch, ONE exchange carries all of them"""
        if self.native:
            return self.ctx.msm_sharded_batch_devptr(self.sid, d_scalars_list, ms)
        return self.fold(self.ctx.msm_batch_devptr(self.sid, d_scalars_list, ms))

    def fold(self, points):
        if self.native:
            return self.ctx.comm_fold(points)
        if self.world == 1 and not self.force_collective:
            return points
        return allgather_fold_many(points, self.device, self.group)


class ShardedProver:
    """prove() with every MSM index-sharded over the ranks (BASELINE configs 4-5: "MSM sharded across 8 x MI355X",
    "the NTT runs single-GPU").  Every rank runs the same prover session on the same witness -- NTTs, grand
    product and quotient are replicated, which costs no communication -- but commits only over its SRS shard;
    the partial commitments of a round are all-gathered and summed in rank order, so every rank feeds identical
    points to its Fiat-Shamir transcript and ends with the identical proof."""

    def __init__(self, sharded: ShardedMsm):
        self.sh = sharded

    def prove(self, circuit: int, wire_evals, pi_evals, cosets, challenge12, challenge34, challenge_v=None):
        return self.sh.ctx.prove(self.sh.sid, circuit, wire_evals, pi_evals, cosets, challenge12, challenge34,
                                 challenge_v=challenge_v, fold=self.sh.fold)

    def prove_native(self, circuit: int, wire_evals, pi_evals, cosets):
        """typlonk_prove on the shard: the library folds every round's commitments itself (needs the native comm)"""
        assert self.sh.native or self.sh.world == 1
        return self.sh.ctx.prove_native(self.sh.sid, circuit, wire_evals, pi_evals, cosets)
