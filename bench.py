#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MSM + NTT hot path on MI355X.

One step = one 2^20-term BLS12-381 G1 MSM (a KZG commit of a 2^20-row wire polynomial,
/root/reference/kzg/src/lib.rs:37-54) with scalars and the SRS already resident in HBM.
With N > 1 ranks (one process per GPU) the base/scalar vectors are index-sharded, each rank runs
its partial MSM and the partial points are combined with one RCCL all-gather + fixed-order fold:
total work is fixed, so scaling is "strong".

  metric  msm_g1_adds_per_s = group operations the kernels EXECUTE for one 2^20-term MSM in the 1-GPU
          configuration / wall time per MSM.  With the fixed-base tables (default, c = 20: 13 windows, one shared
          bucket set) that is 13*m bucket additions + 2*2^19 additions of the row/column bucket reduction
          = 14.7 M; without tables (--tables 0, c = 16) W*m + 2*W*2^(c-1) + c*(W-1) = 17.8 M.  For N > 1 the
          numerator stays the 1-GPU count (total work of the job), so the value moves only with time.
Besides the contract line it reports msm_terms_per_s, the NTT (2^20) time and algorithmic GB/s,
the kernel sequence of one prove() (13 MSM + 15 NTT, plonk/src/proof.rs:96-194) in ms, the
roofline of the dominant kernel (bucket accumulation) from HIP events on the library's stream,
and the reference-faithful CPU MSM timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import typlonk_amd  # noqa: E402
from typlonk_amd.dist import ShardedMsm, local_range  # noqa: E402

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# what actually bounds the accumulation: the vector-ALU rate of the XYZZ mixed addition, measured by the
# self-checking micro-benchmark tools/ubench2 with the shipped flags.  Boxes of the pool differ: 7.13 G/s on the box of
# profiles/r02_ubench2_fused_y3.txt, 7.38-7.46 G/s on the box of profiles/r02_ubench2_sched_variants.txt; the higher
# reading is the ceiling, so that a fast box cannot report a fraction above 1
MIXED_ADD_CEILING = 7.46e9


def fr_mont_limbs(x: int) -> np.ndarray:
    v = (x % FR_MODULUS) * (1 << 256) % FR_MODULUS
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def synthetic_scalars(n: int, seed: int, device) -> torch.Tensor:
    """n valid Fr Montgomery residues (uniform below 2^254 < r), int64 view of the u64 limbs"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
    t[:, 3] &= 0x3FFFFFFFFFFFFFFF
    # the tensor is handed to the library as a raw device pointer (typlonk_*_devptr): finish torch's kernels first --
    # the library reads in the order of ITS stream (typlonk.h, "STREAM ORDERING")
    torch.cuda.synchronize(device)
    return t


def pmc_traffic(kernel: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary
    (profiles/pmc_latest.json = separate --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
    FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; see tools/pmc_summary.py).
    bench.py cannot run the profiler itself, so the value is null when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return float(d[kernel]["traffic_bytes_per_launch"])
    except Exception:
        return None


def prof_ms(ctx, name_prefix: str) -> float:
    return sum(ms for n, ms in ctx.profile() if n.startswith(name_prefix))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=1 << 16,
                    help="terms of the workload timed on the CPU oracle (2^16 ~ 16 s on one host core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--msm-only", action="store_true",
                    help="only the timed MSM loop and its roofline (the command the rocprofv3 summaries "
                         "profiles/r02_kernel_stats_bench_msm_only.csv are taken from)")
    ap.add_argument("--no-sharded-prove", action="store_true",
                    help="N > 1: skip the extra measurement of prove() with every MSM sharded over the ranks")
    ap.add_argument("--tables", type=int, default=20,
                    help="window bits of the fixed-base tables built once per SRS shard (0 = none); used when the "
                         "shard has >= 2^19 points")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # TYPLONK_BENCH_BACKEND=gloo lets several ranks share one GPU (validation of the sharded path on a
    # 1-GPU box: RCCL refuses two ranks on the same device); the exchange then goes through host tensors
    backend = os.environ.get("TYPLONK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or os.environ.get("TYPLONK_FORCE_COLLECTIVE") == "1":
        import datetime

        # a rank that dies must not leave its peers waiting for ever in a collective
        tmo = datetime.timedelta(seconds=int(os.environ.get("TYPLONK_BENCH_PG_TIMEOUT", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    log_n = args.log_n
    n = 1 << log_n
    srs_len = n + 3  # Srs::from_secret(s, gates) has gates + 3 points (kzg/src/srs.rs:31)
    ctx = typlonk_amd.Context(dev_index)
    ctx.set_profiling(True)
    secret = fr_mont_limbs(2)  # the reference's test secret (kzg/src/lib.rs:97)
    sh = ShardedMsm(ctx, srs_len, rank, world, device if backend == "nccl" else torch.device("cpu"))
    sh.generate_srs(secret)
    use_tables = bool(args.tables) and (sh.hi - sh.lo) >= (1 << 16)
    if use_tables:
        ctx.srs_precompute(sh.sid, args.tables)   # setup, like the SRS upload itself: the SRS is fixed per circuit

    full = synthetic_scalars(n, 0x5EED0000 + log_n, device)
    lo, hi = local_range(n, srs_len, world, rank)
    m_local = hi - lo
    c, W, ops_1gpu = ctx.msm_plan(n)
    if bool(args.tables) and srs_len >= (1 << 16):
        # 1-GPU configuration with fixed-base tables: T windows, one shared set of 2^(c-1) buckets
        c, W = args.tables, (256 + args.tables - 1) // args.tables
        ops_1gpu = W * n + 2 * (1 << (c - 1))

    def step():
        return sh.msm_devptr(full.data_ptr(), n)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_ms, accum_launches = {}, 0
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_xy, out_inf = step()
        for name, ms in ctx.profile():
            stage_ms[name] = stage_ms.get(name, 0.0) + ms
            accum_launches += name == "msm_accum"
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = ops_1gpu * args.steps / dt
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}

    result = {
        "metric": "msm_g1_adds_per_s", "value": value, "unit": "G1-adds/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"2^{log_n}-term BLS12-381 G1 MSM (KZG commit of a 2^{log_n}-row polynomial), "
                               f"SRS [s^i]G with s=2, {srs_len} points", "window_bits": c, "windows": W,
                   "parallelism": f"index-sharded x{world} + all-gather fold" if world > 1 else "single GPU",
                   "fixed_base_tables": f"c={args.tables}, {(256 + args.tables - 1) // args.tables} tables" if use_tables else "none"},
        "msm_terms_per_s": n * args.steps / dt,
        "msm_stage_ms": stage_ms,
    }

    # ---- roofline of the dominant kernel (bucket accumulation), HIP events on the launch stream ----
    # A stand-alone MSM of >= 2^20 terms launches the accumulation once per chunk of ~2^19 terms (capi.hip, msm_enqueue):
    # everything below is PER LAUNCH, as rocprofv3's per-kernel average is (profiles/r02_kernel_stats_bench_msm_only.csv).
    launches = max(1, accum_launches // max(1, args.steps))
    t_acc = stage_ms.get("msm_accum", 0.0) * 1e-3 / launches
    terms_per_launch = m_local / launches
    alg_bytes = 128.0 * terms_per_launch  # 32 B scalar + 96 B affine base per term, each read once
    if t_acc > 0:
        ach = alg_bytes / t_acc / 1e9
        adds = W * terms_per_launch / t_acc
        result["roofline"] = {"bound": "hbm", "kernel": "msm_accum_kernel", "achieved": ach, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                              "traffic": pmc_traffic("ty::msm_accum_kernel") if world == 1 and log_n == 20 else None,
                              "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE "
                                                "passes, bytes per launch)",
                              "algorithmic_bytes": alg_bytes, "launches_per_msm": launches,
                              "terms_per_launch": terms_per_launch, "kernel_ms": t_acc * 1e3, "mixed_adds_per_s": adds,
                              "limited_by": "valu",
                              "valu": {"achieved": adds, "peak": MIXED_ADD_CEILING, "unit": "mixed adds/s",
                                       "frac": adds / MIXED_ADD_CEILING,
                                       "peak_source": "tools/ubench2, best box (profiles/r02_ubench2_sched_variants.txt; 7.13 on the box of r02_ubench2_fused_y3.txt)"},
                              "note": "the HBM fraction is what the contract asks for; the kernel is integer-VALU-bound "
                                      "(91 % of the issue slots at 2.06 GHz, profiles/r02_pmc_sq_valu_msm.json) -- see DESIGN.md"}

    if rank == 0 and not args.msm_only:
        # ---- NTT 2^log_n, resident data -----------------------------------------------------------
        v = synthetic_scalars(n, 0xA11CE, device)
        for _ in range(2):
            ctx.ntt_devptr(v.data_ptr(), log_n)
        torch.cuda.synchronize()
        reps, tn, kern = 10, 0.0, 0.0
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.ntt_devptr(v.data_ptr(), log_n, inverse=bool(_ & 1))
            kern += prof_ms(ctx, "ntt_")
        torch.cuda.synchronize()
        tn = (time.perf_counter() - t1) / reps
        result["ntt"] = {"log_n": log_n, "ms": tn * 1e3, "kernel_ms": kern / reps,
                         "algorithmic_GBps": 64.0 * n / (kern / reps * 1e-3) / 1e9,
                         "frac_of_hbm_peak": 64.0 * n / (kern / reps * 1e-3) / 1e9 / HBM_PEAK_GBS}
        try:   # PMC traffic of one pass (two passes per 2^20 transform, three at 2^22), profiles/r02_pmc_ntt.json
            with open(os.path.join(ROOT, "profiles", "r02_pmc_ntt.json")) as f:
                pm = json.load(f).get(f"ntt_pass_kernel n=2^{log_n}")
            if pm:
                result["ntt"]["pmc_traffic_bytes_per_pass"] = pm["traffic_bytes_per_pass"]
                result["ntt"]["note"] = ("VALU-bound (80 % Fr multiplications, DESIGN.md section 5): the algorithmic figure "
                                         "counts 64 B per element once; the two passes of a 2^20 transform move ~2.8x that")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.msm_only:
        # ---- kernel sequence of one prove(): 15 size-n NTTs, the quotient on the
        # 4n coset domain (per-circuit constants cached by typlonk_circuit_load), 13 MSMs in the groups
        # prove() issues them.  Timing only: the polynomials are random, not a satisfying witness.
        bufs = [ctx.alloc(n) for _ in range(13)]
        for i, bf in enumerate(bufs):
            bf.upload(synthetic_scalars(n, 0xB0B + i, device).cpu().numpy().view(np.uint64))
        wires, zbuf, pibuf, sel, sig = bufs[0:3], bufs[3], bufs[4], bufs[5:10], bufs[10:13]
        t_out = ctx.alloc(4 * n)
        cid = ctx.circuit_load(log_n, sel, sig)
        one = fr_mont_limbs(1)
        chal = [fr_mont_limbs(0x1234567 + k) for k in range(3)]
        cosets = [fr_mont_limbs(k) for k in (2, 3, 4)]
        ptr = [bf.devptr for bf in bufs[:6]]

        def prove_sequence(batched: bool):
            tq = 0.0
            for i in range(15):
                ctx.ntt_devptr(ptr[i % 3], log_n, inverse=(i >= 3))
            groups = [[(ptr[0], n), (ptr[1], n), (ptr[2], n)], [(ptr[3], n)]]
            groups.append([(ptr[k % 6], n - 1) for k in range(6)])                       # openings
            groups.append([(t_out.devptr, n), (t_out.devptr + 32 * n, n), (t_out.devptr + 64 * n, n - 3)])
            for gi, g in enumerate(groups):
                if gi == 2:  # the quotient is built after Z is committed (proof.rs:139-145)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    ctx.quotient_dev(log_n, wires, zbuf, None, None, pibuf, chal[0], chal[1], chal[2], cosets, t_out,
                                     circuit=cid)
                    torch.cuda.synchronize()
                    tq = (time.perf_counter() - t2) * 1e3
                if batched:
                    ctx.msm_batch_devptr(sh.sid, [p for p, _ in g], [mm for _, mm in g])
                else:
                    for p, mm in g:
                        ctx.msm_devptr(sh.sid, p, mm)
            torch.cuda.synchronize()
            return tq

        for batched, key in ((False, "prove_hotpath_sequential_ms"), (True, "prove_hotpath_ms")):
            prove_sequence(batched)  # warm-up (allocates workspaces / tables on first use)
            t1 = time.perf_counter()
            tq = prove_sequence(batched)
            result[key] = (time.perf_counter() - t1) * 1e3
            result["quotient_ms"] = tq
        ctx.circuit_free(cid)
        for bf in bufs + [t_out]:
            bf.free()

        # ---- a real prove(): squaring-chain circuit with n = 2^log_n rows, satisfying witness, the whole
        # device-side flow of plonk::proof::prove (rounds 1-3; Fiat-Shamir challenges injected) ---------------
        from typlonk_amd.circuits import SquaringChain

        chain = SquaringChain(ctx, log_n)
        ch = [fr_mont_limbs(0x1234567 + k) for k in range(4)]
        zero_limbs = np.zeros(4, dtype=np.uint64)

        def run_prove():
            return ctx.prove(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,
                             lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))

        run_prove()
        torch.cuda.synchronize()
        reps = 3
        t1 = time.perf_counter()
        for _ in range(reps):
            proof = run_prove()
        torch.cuda.synchronize()
        result["prove_ms"] = (time.perf_counter() - t1) / reps * 1e3
        result["prove_valid"] = bool((proof["evals"][5] == zero_limbs).all())   # r(zeta) == 0, proof.rs:234-235
        result["prove_config"] = f"squaring chain, {chain.gates} gates, n = 2^{log_n}, 7 commitments + 6 openings"
        # same proof with the openings at zeta batched (typlonk_prover_round3_evals / round4_batched): 9 MSMs
        run_b = lambda: ctx.prove(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,   # noqa: E731
                                  lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]), challenge_v=lambda e: ch[1])
        run_b()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            run_b()
        torch.cuda.synchronize()
        result["prove_batched_openings_ms"] = (time.perf_counter() - t1) / reps * 1e3
        # the same proof through the one-call native entry point (typlonk_prove: the reference's own Fiat-Shamir
        # transcript, restated natively, between the rounds -- no Python inside the proof)
        ctx.prove_native(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            pn = ctx.prove_native(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        torch.cuda.synchronize()
        result["prove_native_ms"] = (time.perf_counter() - t1) / reps * 1e3
        result["prove_native_valid"] = bool((pn["evals"][5] == zero_limbs).all())
        chain.free()

        if not args.no_cpu_baseline:
            # ---- parity gate + CPU baseline: the oracle is the checker, timed on a bounded sample --
            from oracle import coracle as CO

            ms = min(args.cpu_sample, n)
            sc_host = full[:ms].cpu().numpy().view(np.uint64)
            xy, inf = ctx.srs_download(sh.sid, 0, ms)
            CO.group_ops_reset()
            t1 = time.perf_counter()
            ref_xy, ref_inf = CO.msm_reference(sc_host, xy, inf)
            tc = time.perf_counter() - t1
            cpu_ops = CO.group_ops()
            gxy, ginf = ctx.msm_devptr(sh.sid, full.data_ptr(), ms)
            parity_sample = bool((gxy == ref_xy).all() and ginf == ref_inf)
            # full size: commit(p) == [p(s)]G  (the reference's own test identity, kzg/src/lib.rs:102-105)
            ps = CO.poly_eval(full.cpu().numpy().view(np.uint64), secret)
            exp_xy, exp_inf = CO.g1_mul_generator(ps)
            parity_full = bool((np.asarray(out_xy) == exp_xy).all() and out_inf == exp_inf)
            result["parity"] = {"sample_vs_oracle": parity_sample, "full_commit_identity": parity_full}
            result["cpu_baseline"] = {
                "value": cpu_ops / tc, "unit": "G1-adds/s", "cores": 1, "kind": "port",
                "sample": f"first {ms} terms of the same scalar/SRS vectors, reference-faithful per-term "
                          f"double-and-add + affine normalisation + sum (kzg/src/lib.rs:41-54)",
                "terms_per_s": ms / tc, "seconds": tc, "host_cpus": os.cpu_count()}
            # "fair CPU" (SURVEY 8d-ii): bucket method on all host cores (oracle_msm_pippenger) + the radix-2 NTT
            mf = min(1 << 18, n)
            scf = full[:mf].cpu().numpy().view(np.uint64)
            xyf, inff = ctx.srs_download(sh.sid, 0, mf)
            cf = max(4, min(16, mf.bit_length() - 4))
            CO.msm_pippenger(scf[:1024], xyf[:1024], inff[:1024], c=8)      # spin the OpenMP team up
            t1 = time.perf_counter()
            f_xy, f_inf, f_ops, f_thr = CO.msm_pippenger(scf, xyf, inff, c=cf)
            tf = time.perf_counter() - t1
            gxy, ginf = ctx.msm_devptr(sh.sid, full.data_ptr(), mf)
            parity_fair = bool((gxy == f_xy).all() and ginf == f_inf)
            t1 = time.perf_counter()
            CO.ntt(scf, mf.bit_length() - 1)
            tn = time.perf_counter() - t1
            result["parity"]["sample_vs_fair_cpu"] = parity_fair
            result["cpu_fair"] = {
                "kind": "port (bucket method, not the reference's algorithm)", "cores": f_thr, "window_bits": cf,
                "sample": f"first {mf} terms", "seconds": tf, "terms_per_s": mf / tf, "value": f_ops / tf,
                "unit": "G1-adds/s", "ntt_ms_1_thread": tn * 1e3, "ntt_log_n": mf.bit_length() - 1}
            # the north-star's ">= 10x prove() over the CPU reference": prove()'s 13 MSMs alone (proof.rs call sites, SURVEY
            # 8a7) at the two measured CPU rates -- a LOWER bound of the CPU time (its NTTs, and the reference's schoolbook
            # quotient of ~19 n^2 multiplications, are not counted)
            if "prove_ms" in result:
                msm_terms = 13 * n
                result["prove_vs_cpu"] = {
                    "gpu_prove_ms": result["prove_ms"],
                    "cpu_reference_path_13_msms_s": msm_terms / (ms / tc), "cpu_reference_cores": 1,
                    "cpu_all_cores_bucket_method_13_msms_s": msm_terms / (mf / tf), "cpu_all_cores": f_thr,
                    "speedup_vs_reference_path": msm_terms / (ms / tc) / (result["prove_ms"] * 1e-3),
                    "speedup_vs_all_cores": msm_terms / (mf / tf) / (result["prove_ms"] * 1e-3),
                    "note": "CPU side = MSMs only, extrapolated from the timed samples; a lower bound of the CPU prove()"}
            if not (parity_sample and parity_full and parity_fair):
                result["value"] = None
                result["error"] = "GPU result differs from the oracle: number withheld"

    if world > 1 and not args.no_sharded_prove and not args.msm_only:
        # ---- prove() with every MSM sharded over the ranks (NTT / quotient replicated): typlonk_amd.dist.ShardedProver.
        # The block contains collectives, so the ranks AGREE on success before entering it and after it: a failure on
        # one rank (OOM, HIP error) must not leave its peers blocked in an all-gather.  A rank that fails inside the
        # timed proofs raises (its peers' collectives then end with the process-group timeout); nothing is swallowed.
        def agree(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        from typlonk_amd.circuits import SquaringChain
        from typlonk_amd.dist import ShardedProver

        chain, setup_err = None, None
        try:
            chain = SquaringChain(ctx, log_n)
        except Exception as e:  # setup only: no collective has been entered yet
            setup_err = f"{type(e).__name__}: {e}"
        if agree(setup_err is None):
            ch = [fr_mont_limbs(0x1234567 + k) for k in range(4)]
            sp = ShardedProver(sh)
            run_s = lambda: sp.prove(chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,   # noqa: E731
                                     lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]), challenge_v=lambda e: ch[1])
            run_s()
            sync_all()
            t1 = time.perf_counter()
            for _ in range(3):
                proof = run_s()
            sync_all()
            result["prove_sharded_batched_ms"] = (time.perf_counter() - t1) / 3 * 1e3
            result["prove_valid"] = bool((proof["evals"][5] == np.zeros(4, dtype=np.uint64)).all())
        else:
            result["prove_sharded_error"] = setup_err or "setup failed on another rank"
        if chain is not None:
            chain.free()

    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        # sharded result vs the reference's own test identity commit(p) == [p(s)]G (oracle = checker)
        from oracle import coracle as CO

        ps = CO.poly_eval(full.cpu().numpy().view(np.uint64), secret)
        exp_xy, exp_inf = CO.g1_mul_generator(ps)
        ok = bool((np.asarray(out_xy) == exp_xy).all() and out_inf == exp_inf)
        result["parity"] = {"full_commit_identity": ok}
        if not ok:
            result["value"] = None
            result["error"] = "sharded GPU result differs from the oracle: number withheld"
    # everything that can still print (RCCL reports its library path when a communicator comes up or goes away) happens
    # before the contract line, so that the JSON is the last line of rank 0's output
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    # librccl prints through C stdio, which is block-buffered when stdout is a pipe or a file and would otherwise be
    # flushed at exit, after Python's own line
    import ctypes

    ctypes.CDLL(None).fflush(None)
    sys.stderr.flush()
    if rank == 0:
        if world > 1:
            time.sleep(0.5)  # the other ranks share this stdout: let their last lines land first
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
