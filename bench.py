#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MSM + NTT hot path on MI355X.

One step = one 2^20-term BLS12-381 G1 MSM (a KZG commit of a 2^20-row wire polynomial,
/root/reference/kzg/src/lib.rs:37-54) with scalars and the SRS already resident in HBM.
With N > 1 ranks (one process per GPU) the base/scalar vectors are index-sharded, each rank runs
its partial MSM and the partial points are combined with one RCCL all-gather + fixed-order fold
inside the library (typlonk_msm_g1_sharded_devptr): total work is fixed, so scaling is "strong".

  metric  msm_g1_adds_per_s = group operations the kernels EXECUTE for one 2^20-term MSM in the 1-GPU
          configuration / wall time per MSM.  With the fixed-base tables (default, c = 20: 13 windows, one shared
          bucket set) that is 13*m bucket additions + 2*2^19 additions of the row/column bucket reduction
          = 14.7 M; without tables (--tables none, c = 16) W*m + 2*W*2^(c-1) + c*(W-1) = 17.8 M.  For N > 1 the
          numerator stays the 1-GPU count (total work of the job), so the value moves only with time.
Besides the contract line it reports msm_terms_per_s, the NTT (2^20) time with its algorithmic GB/s and its
Fr-multiplication rate against the instruction-count ceiling, the kernel sequence of one prove() (13 MSM + 15 NTT,
plonk/src/proof.rs:96-194) in ms, a real prove(), the roofline of the dominant kernel (bucket accumulation) from HIP
events on the library's stream, and the CPU baselines of BASELINE.md section 3 timed on this box's host cores.

The contract line is printed from a `finally`: a failure in any later section (or in the multi-GPU teardown) is
recorded in the line (`*_error` keys) instead of losing the numbers already measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

import typlonk_amd  # noqa: E402
from typlonk_amd.dist import ShardedMsm, local_range  # noqa: E402

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


# What actually bounds the accumulation (and the NTT) is the vector ALU.  The ceilings are INSTRUCTION COUNTS x a price
# list x the chip's maximum clock, not timings of the same loops: the steady-state path of msm_accum_kernel's inner loop
# holds 3081 v_mad_u64_u32 per mixed addition (4818 VALU instructions in all; `hipcc -S` listing read by
# tools/isa_count.py -> profiles/r04_isa_msm_accum.json).  The price list is round 4's (tools/ubench4.hip,
# profiles/r04_ubench4_pricing.txt): cycles per wave-instruction per SIMD at saturation, counted with s_memtime per
# physical SIMD over an occupancy sweep -- 2.20 for full-rate VOP2 (the guide's 2-cycle wave64 issue), 4.12 for the
# half-rate class, 4.27 for v_mad_u64_u32 and the carry ops.  At the 2400 MHz maximum clock and 1024 SIMDs x 64 lanes:
# 11.96 G mixed additions/s if only the multiplier instructions issued, 8.40 G/s for all 4818 instructions (18.7 k cycles
# per wavefront and addition; the whole-addition micro-kernel of the same file measures 19.5 k).  The kernel itself is
# power-limited to 2.06-2.2 GHz: that shortfall is part of the distance to either ceiling.
def _isa(name: str, key: str, default):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)[key]
    except Exception:
        return default


def _sq(kind: str, key: str) -> dict:
    """SQ-counter summary of a kernel from the newest committed profiles/r0x_pmc_sq_valu_<kind>.json that has it"""
    for rnd in ("r06", "r05"):
        d = _isa(f"{rnd}_pmc_sq_valu_{kind}.json", key, None)
        if d:
            return dict(d, source=f"profiles/{rnd}_pmc_sq_valu_{kind}.json")
    return {}


MIXED_ADD_MULTIPLIER_CEILING = float(_isa("r04_isa_msm_accum.json", "ceiling_units_per_s_multiplier_only", 11.956e9))
MIXED_ADD_ALL_VALU_MODEL = float(_isa("r04_isa_msm_accum.json", "ceiling_units_per_s_all_valu", 8.398e9))
# the same for one Fr multiplication inside the NTT butterflies.  Since round 4 the default kernel is the one on nine 30-bit
# limbs (profiles/r04_isa_ntt30.json: 153.75 v_mad_u64_u32 per multiplication, 1155 VALU instructions per radix-4 group of
# four multiplications = 4156 cycles; the 8 x 32 kernel, profiles/r04_isa_ntt.json: 128 + a carry op each, 5655 cycles)
FR_MUL_MULTIPLIER_CEILING = float(_isa("r04_isa_ntt30.json", "ceiling_fr_mul_per_s_multiplier_only", 2.396e11))
FR_MUL_ALL_VALU_MODEL = float(_isa("r04_isa_ntt30.json", "ceiling_fr_mul_per_s_all_valu", 1.514e11))


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


def table_windows(c: int) -> int:
    """windows of the table-mode MSM with c-bit windows (launch.hpp msm_windows: centred scalars where that saves one)"""
    def w(bits):
        w0 = (bits + c - 1) // c
        return w0 + (1 if bits - c * (w0 - 1) == c else 0)
    return min(w(255), w(254))


class Section:
    """`with Section(result, "ntt"):` -- an exception inside is recorded as result["ntt_error"] and does not stop the run"""

    def __init__(self, result, name):
        self.result, self.name = result, name

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception):
            self.result[self.name + "_error"] = f"{et.__name__}: {ev}"
            traceback.print_exception(et, ev, tb, file=sys.stderr)
            return True
        return False


# ---- N > 1: what the line is read against (review of round 4) -----------------------------------------------------------
# A scaling record of this bench has three modes next to each other -- ONE stand-alone MSM per step (`value`, the headline),
# nine MSMs in flight per call (`msm_batch`, how prove() issues its commitments) and the whole sharded prove() -- and the
# north-star's ">= 6x at 8 GPUs" is claimed for the batched mode only (a stand-alone 2^17-term MSM has a
# latency floor of ~0.54 ms -> ~4.5-4.8x).  So that a record can be read without re-deriving that, every N > 1 line
# carries (a) the one-GPU figures of this round it should be divided into, from the committed one-GPU record, and (b) the
# projection made on ONE GPU acting as rank 0 of N (tools/shard_latency.py: real kernels on a 1/N shard, the exchange on a
# one-rank communicator -- launch, copy and synchronisation latencies but no wire time).
def _latest_profile(suffix: str):
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(n for n in os.listdir(pdir) if n.endswith(suffix) and n.startswith("r0"))
    except OSError:
        return None
    return os.path.join(pdir, names[-1]) if names else None


def one_gpu_reference(log_n: int):
    path = _latest_profile("_bench_full.json" if log_n == 20 else f"_bench_2_{log_n}.json")
    if not path:
        return None
    try:
        d = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None
    return {"ms_per_step": d.get("ms_per_step"), "msm_batch_ms_per_msm": (d.get("msm_batch") or {}).get("ms_per_msm"),
            "prove_ms": d.get("prove_native_ms") or d.get("prove_ms"), "prove_batched_openings_ms": d.get("prove_batched_openings_ms"),
            "source": os.path.relpath(path, ROOT) + " (one MI355X, same code; another box of the pool)"}


def expected_from_1gpu(log_n: int, world: int, ref):
    path = _latest_profile("_shard_latency.jsonl")
    if not path or log_n != 20:
        return None
    row = None
    for line in open(path):
        if line.startswith("SHARD "):
            d = json.loads(line[6:])
            if d.get("world") == world and d.get("log_n") == log_n:
                row = d
    if not row:
        return None
    # (the *_projected figures include the host fold over N ranks' records, which a one-rank communicator does not show)
    row = dict(row)
    row["sharded_msm_wall_ms"] = row.get("sharded_msm_wall_projected_ms") or row.get("sharded_msm_wall_ms")
    row["sharded_batch9_ms_per_msm"] = row.get("sharded_batch9_ms_per_msm_projected") or row.get("sharded_batch9_ms_per_msm")
    out = {"one_msm_plus_exchange_ms": row.get("sharded_msm_wall_ms"), "batched_ms_per_msm": row.get("sharded_batch9_ms_per_msm"),
           "prove_on_shard_ms": row.get("prove_on_shard_ms"), "source": os.path.relpath(path, ROOT) +
           " (tools/shard_latency.py: one GPU as rank 0 of N, exchange on a one-rank communicator)"}
    if ref:
        sp = {}
        for k_ref, k_row, name in (("ms_per_step", "sharded_msm_wall_ms", "one_msm"), ("msm_batch_ms_per_msm", "sharded_batch9_ms_per_msm", "batched_msms"),
                                   ("prove_ms", "prove_on_shard_ms", "prove")):
            if ref.get(k_ref) and row.get(k_row):
                sp[name] = round(ref[k_ref] / row[k_row], 2)
        out["speedup"] = sp
    return out


def bench_one_gpu_same_run(dev_index: int, log_n: int, device, result) -> None:
    """Rank 0 alone, after the sharded measurements: the UNSHARDED MSM of the same size on its own GPU -- a second context
    with the full SRS and the one-GPU tables -- stand-alone and nine in flight.  The N-GPU figures of this line divided by
    THESE are ratios of one box and one run (`scaling_same_run`); the committed one-GPU record of another box is context
    only (`vs_committed_reference`)."""
    n = 1 << log_n
    c1 = typlonk_amd.Context(dev_index)
    try:
        sid = c1.srs_generate(fr_mont_limbs(2), n + 3)
        c1.srs_precompute(sid, 20 if log_n >= 19 else 0)
        sc = synthetic_scalars(n, 0x5EED0000 + log_n, device)
        for _ in range(3):
            c1.msm_devptr(sid, sc.data_ptr(), n)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            c1.msm_devptr(sid, sc.data_ptr(), n)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / reps * 1e3
        ptrs, ms = [sc.data_ptr()] * 9, [n - (k % 3) for k in range(9)]
        c1.msm_batch_devptr(sid, ptrs, ms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            c1.msm_batch_devptr(sid, ptrs, ms)
        torch.cuda.synchronize()
        bat = (time.perf_counter() - t0) / 3 / 9 * 1e3
        result["one_gpu_same_run"] = {"ms_per_step": one, "msm_batch_ms_per_msm": bat,
                                      "note": "rank 0 alone, same process and box, after the sharded measurements"}
        sr = {}
        if result.get("ms_per_step"):
            sr["standalone"] = round(one / result["ms_per_step"], 3)
        mb = (result.get("msm_batch") or {}).get("ms_per_msm")
        if mb:
            sr["batched"] = round(bat / mb, 3)
        result["scaling_same_run"] = sr
    finally:
        c1.close()


def add_scaling_context(result, log_n: int, world: int) -> None:
    ref = one_gpu_reference(log_n)
    if ref:
        result["one_gpu_reference"] = ref
        sc = {"mode_of_value": "one stand-alone MSM per step",
              "read_as": "CONTEXT ONLY: this run's timings over the committed one-GPU record of ANOTHER box and run (boxes of the pool "
                         "differ by +-3 %); the same-run ratios are in scaling_same_run",
              "claim": "the north-star's >= 6x at 8 GPUs is claimed for `batched` (nine MSMs in flight per call), not for `standalone`"}
        if ref.get("ms_per_step") and result.get("ms_per_step"):
            sc["standalone"] = round(ref["ms_per_step"] / result["ms_per_step"], 3)
        mb = (result.get("msm_batch") or {}).get("ms_per_msm")
        if ref.get("msm_batch_ms_per_msm") and mb:
            sc["batched"] = round(ref["msm_batch_ms_per_msm"] / mb, 3)
        for key, name, rk in (("prove_sharded_native_ms", "prove_native", "prove_ms"), ("prove_sharded_ms", "prove", "prove_ms"),
                              ("prove_sharded_batched_ms", "prove_batched_openings", "prove_batched_openings_ms")):
            if ref.get(rk) and result.get(key):
                sc[name] = round(ref[rk] / result[key], 3)
        result["vs_committed_reference"] = sc
    exp = expected_from_1gpu(log_n, world, ref)
    if exp:
        result["expected_from_1gpu"] = exp


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes of this same script (the
    environment torch.distributed.run would give them: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT),
    relay rank 0's contract line as the LAST line of stdout and exit with the ranks' code.  The parent never initialises
    the GPU (torch.cuda.device_count() does not, on this image) and never exec()s.  With fewer devices than ranks the
    ranks share devices over gloo (RCCL refuses two ranks per device): the line then says so (`ranks_share_gpu`) --
    a validation of the sharded path, not a scaling number.  A failing rank still leaves a contract line with "error"."""
    import socket
    import subprocess
    import threading

    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    ndev = torch.cuda.device_count()
    if ndev < n and "TYPLONK_BENCH_BACKEND" not in env:
        env["TYPLONK_BENCH_BACKEND"] = "gloo"
    if ndev < n:
        env["TYPLONK_BENCH_SHARED_GPU"] = f"{n} ranks on {ndev} device(s)"
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    last = [None]

    def relay():
        for line in procs[0].stdout:
            if line.startswith('{"metric"'):
                last[0] = line.rstrip("\n")          # held back: it must be the last line of the parent's stdout
            else:
                sys.stdout.write(line)
                sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    # a rank that dies leaves its peers inside a collective until the process-group timeout: give them a grace period,
    # then end exactly the processes started here
    deadline = None
    while any(p.poll() is None for p in procs):
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.time() + float(os.environ.get("TYPLONK_BENCH_GRACE_S", "30"))
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    th.join(timeout=10)
    codes = [p.returncode for p in procs]
    code = next((c for c in codes if c), 0)
    if last[0] is not None:
        line = last[0]
        if code and '"error"' not in line:
            d = json.loads(line)
            d["error"] = f"rank exit codes {codes}"
            line = json.dumps(d)
        print(line, flush=True)
    else:
        print(json.dumps({"metric": "msm_g1_adds_per_s", "value": None, "unit": "G1-adds/s", "n_gpus": n,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                          "error": f"no contract line from rank 0; rank exit codes {codes}"}), flush=True)
        code = code or 1
    return code if code > 0 else 1 if code else 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--warm-seconds", type=float, default=1.0,
                    help="untimed steps after the --warmup steps until this much wall time has passed (clock ramp of a fresh box); 0 = none")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=1 << 16,
                    help="terms of the workload timed on the reference-faithful CPU MSM (2^16 ~ 16 s on one host core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true",
                    help="the long CPU legs of BASELINE.md section 3 as well: reference-faithful MSM on the full vector "
                         "(minutes on one core), schoolbook quotient at 2^12")
    ap.add_argument("--msm-only", action="store_true",
                    help="only the timed MSM loop and its roofline (the command the rocprofv3 summaries "
                         "profiles/r0x_kernel_stats_bench_msm_only.csv are taken from)")
    ap.add_argument("--no-prove-2-22", action="store_true",
                    help="skip the extra 2^22-row prove() (BASELINE config 5's size) the default 2^20 run reports next to the headline")
    ap.add_argument("--no-sharded-prove", action="store_true",
                    help="N > 1: skip the extra measurement of prove() with every MSM sharded over the ranks")
    ap.add_argument("--tables", default="auto",
                    help="fixed-base tables built once per SRS (shard): window bits 14..20, 'none', or 'auto' = 20 on one "
                         "GPU, the library's choice by shard length (15 below 2^16 points, 17 below 2^19) on several")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher (it never touches the GPU)
        sys.exit(self_launch(args))
    if world != args.gpus:
        print(json.dumps({"metric": "msm_g1_adds_per_s", "value": None, "unit": "G1-adds/s", "n_gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                          "error": f"--gpus {args.gpus} but WORLD_SIZE={world}"}), flush=True)
        sys.exit(2)
    # TYPLONK_BENCH_BACKEND=gloo lets several ranks share one GPU (validation of the sharded path on a
    # 1-GPU box: RCCL refuses two ranks on the same device); the exchange then goes through host tensors
    backend = os.environ.get("TYPLONK_BENCH_BACKEND", "nccl")
    result: dict = {"metric": "msm_g1_adds_per_s", "value": None, "unit": "G1-adds/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                    "vs_baseline": None, "dtype": "u32", "data": "synthetic"}
    state: dict = {"ctx": None}
    code = 0
    try:
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
        run(args, rank, world, backend, dev_index, device, result, state)
    except BaseException as e:  # noqa: BLE001 -- the line below must still be printed
        result["error"] = f"{type(e).__name__}: {e}"
        traceback.print_exc(file=sys.stderr)
        code = 1
    finally:
        # Everything that can still print (RCCL reports its library path when a communicator comes up or goes away)
        # happens before the contract line, so that the JSON is the last line of rank 0's output; nothing here may
        # prevent it from appearing.
        try:
            if dist.is_initialized():
                if code == 0:
                    dist.barrier()
                dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001
            result["teardown_error"] = f"{type(e).__name__}: {e}"
        try:
            if state["ctx"] is not None:
                state["ctx"].close()
        except Exception as e:  # noqa: BLE001
            result["teardown_error"] = f"{type(e).__name__}: {e}"
        # librccl prints through C stdio, which is block-buffered when stdout is a pipe or a file and would otherwise
        # be flushed at exit, after Python's own line
        import ctypes

        ctypes.CDLL(None).fflush(None)
        sys.stderr.flush()
        if rank == 0:
            if world > 1:
                time.sleep(0.5)  # the other ranks share this stdout: let their last lines land first
            print(json.dumps(result), flush=True)
    if code:
        sys.exit(code)


def run(args, rank, world, backend, dev_index, device, result, state) -> None:
    log_n = args.log_n
    n = 1 << log_n
    srs_len = n + 3  # Srs::from_secret(s, gates) has gates + 3 points (kzg/src/srs.rs:31)
    ctx = typlonk_amd.Context(dev_index)
    state["ctx"] = ctx
    # the timed region brackets ONLY the dominant kernel with HIP events (level 2: the roofline's `achieved` is measured
    # live over the timed steps); the full stage split is collected afterwards, untimed -- bracketing every stage costs
    # ~0.1 ms of event traffic per MSM, which a caller never pays
    ctx.set_profiling(2)
    secret = fr_mont_limbs(2)  # the reference's test secret (kzg/src/lib.rs:97)
    sh = ShardedMsm(ctx, srs_len, rank, world, device if backend == "nccl" else torch.device("cpu"))
    sh.generate_srs(secret)
    # what the record must show about the exchange: the size of the RCCL communicator the LIBRARY holds (typlonk_comm_info;
    # 0 = no native communicator: one rank, or the torch.distributed fallback named in `exchange`)
    result["rccl_world"] = int(ctx.comm_info()[1]) if sh.native else 0
    result["exchange"] = ("rccl all-gather inside the library" if sh.native else
                          "none (one rank)" if world == 1 and not sh.force_collective else f"torch.distributed {backend}")
    if os.environ.get("TYPLONK_BENCH_SHARED_GPU"):
        result["ranks_share_gpu"] = os.environ["TYPLONK_BENCH_SHARED_GPU"] + ": validation of the sharded path, not a scaling figure"
    # setup, like the SRS upload itself: the SRS is fixed per circuit (plonk/src/lib.rs:22)
    tables_c = None
    if args.tables != "none" and (sh.hi - sh.lo) >= (1 << 16):
        tables_c = (20 if world == 1 else 0) if args.tables == "auto" else int(args.tables)
        ctx.srs_precompute(sh.sid, tables_c)
        if tables_c == 0:
            tables_c = 17 if (sh.hi - sh.lo) < (1 << 19) else 20

    full = synthetic_scalars(n, 0x5EED0000 + log_n, device)
    lo, hi = local_range(n, srs_len, world, rank)
    m_local = hi - lo
    # the numerator: group operations of the 1-GPU configuration
    c1, W1, ops_1gpu = ctx.msm_plan(n)
    if args.tables != "none" and srs_len >= (1 << 16):
        c1 = 20 if args.tables == "auto" else int(args.tables)
        W1 = table_windows(c1)
        ops_1gpu = W1 * n + 2 * (1 << (c1 - 1))   # T windows into one shared set of 2^(c-1) buckets + its reduction

    def step():
        return sh.msm_devptr(full.data_ptr(), n)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ... and, on top of the W warm-up steps, untimed steps until the device has been busy for --warm-seconds: a fresh box
    # (the first program after the lease) runs the first ~0.5 s of kernels at a lower clock -- the same MSM reads 2.51 ms
    # as the first thing on a box and 2.40 ms after other work (profiles/r06_bench_full.json vs r06_bench_final_tree.json)
    # (a COUNT, the same on every rank -- the steps of a sharded run contain collectives: rank 0's estimate is broadcast)
    warm_extra = 0
    if args.warm_seconds > 0:
        torch.cuda.synchronize()
        t_e = time.perf_counter()
        for _ in range(3):                     # (three steps to price one, after the allocations of the first warm-up steps)
            step()
        est = max((time.perf_counter() - t_e) / 3, 1e-4)
        warm_extra = min(2000, int(args.warm_seconds / est))
    if world > 1:
        t = torch.tensor([warm_extra], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        dist.broadcast(t, src=0)
        warm_extra = int(t.item())
    for _ in range(warm_extra):
        step()
    result["warm_extra_steps"] = warm_extra
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
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    # the full stage split (sort / accumulate / reduce): a few untimed steps with every stage bracketed
    ctx.set_profiling(1)
    extra, full_ms = 5, {}
    for _ in range(extra):
        step()
        for name, ms in ctx.profile():
            full_ms[name] = full_ms.get(name, 0.0) + ms / extra
    for name, ms in full_ms.items():
        if name != "msm_accum":
            stage_ms[name] = ms
    stage_ms_note = ("msm_accum: HIP events over the timed steps; the other stages: five untimed steps with every stage bracketed "
                     "(typlonk_set_profiling 2 / 1)")
    Wl = table_windows(tables_c) if tables_c else ctx.msm_plan(max(m_local, 1))[1]
    result.update({
        "value": ops_1gpu * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
        "config": {"workload": f"2^{log_n}-term BLS12-381 G1 MSM (KZG commit of a 2^{log_n}-row polynomial), "
                               f"SRS [s^i]G with s=2, {srs_len} points", "window_bits": c1, "windows": W1,
                   "parallelism": (f"index-sharded x{world}, RCCL all-gather + fold inside the library" if sh.native else
                                   f"index-sharded x{world} + all-gather fold ({backend})") if world > 1 else "single GPU",
                   "fixed_base_tables": f"c={tables_c}, {Wl} tables per rank" if tables_c else "none"},
        "msm_terms_per_s": n * args.steps / dt,
        "msm_stage_ms": stage_ms, "msm_stage_ms_note": stage_ms_note,
    })

    # ---- a prover round's worth of MSMs in one call (typlonk_msm_g1_batch_devptr / _sharded_batch_devptr): sort and
    # reduction of one MSM hide behind the accumulation of another, and N ranks exchange all nine sums at once.  Not the
    # headline (one MSM at a time, above), reported next to it because this is how prove() issues its commitments.
    if not args.msm_only:
        try:
            nb = 9
            ptrs, ms = [full.data_ptr()] * nb, [n - (k % 3) for k in range(nb)]
            sh.msm_batch_devptr(ptrs, ms)
            sync_all()
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                outs = sh.msm_batch_devptr(ptrs, ms)
            sync_all()
            tb = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([tb], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tb = float(t.item())
            same = bool((np.asarray(outs[0][0]) == np.asarray(out_xy)).all() and outs[0][1] == out_inf)
            result["msm_batch"] = {"msms": nb, "ms_per_msm": tb / reps / nb * 1e3, "first_equals_single": same}
        except Exception as e:  # noqa: BLE001 -- a secondary figure must not cost the headline
            if world > 1:
                raise                     # inside a collective sequence: let the ranks fail together
            result["msm_batch_error"] = f"{type(e).__name__}: {e}"

    # ---- roofline of the dominant kernel (bucket accumulation), HIP events on the launch stream ----
    # A stand-alone MSM of >= 2^20 terms launches the accumulation once per chunk of ~2^19 terms (msm_host.hip, msm_enqueue):
    # everything below is PER LAUNCH, as rocprofv3's per-kernel average is (profiles/r0x_kernel_stats_bench_msm_only.csv).
    launches = max(1, accum_launches // max(1, args.steps))
    t_acc = stage_ms.get("msm_accum", 0.0) * 1e-3 / launches
    terms_per_launch = m_local / launches
    alg_bytes = 128.0 * terms_per_launch  # 32 B scalar + 96 B affine base per term, each read once
    if t_acc > 0:
        ach = alg_bytes / t_acc / 1e9
        adds = Wl * terms_per_launch / t_acc
        traffic = pmc_traffic("ty::msm_accum_kernel") if world == 1 and log_n == 20 else None
        result["roofline"] = {
            "bound": "valu", "kernel": "msm_accum_kernel" if world == 1 else "msm_accum_kernel / msm_accum_ml_kernel",
            "achieved": adds, "peak": MIXED_ADD_MULTIPLIER_CEILING, "unit": "mixed adds/s",
            "frac": adds / MIXED_ADD_MULTIPLIER_CEILING, "traffic": traffic,
            "peak_source": "instruction count: 3081 v_mad_u64_u32 per mixed addition on the loop's steady-state path "
                           "(profiles/r04_isa_msm_accum.json) x 4.27 cycles per v_mad_u64_u32 and SIMD at saturation (s_memtime, "
                           "occupancy sweep: profiles/r04_ubench4_pricing.txt) at the 2400 MHz maximum clock x 1024 SIMDs x 64 lanes "
                           "-- multiplier instructions only; reproducible by hand from that one file",
            "all_valu_model": {"peak": MIXED_ADD_ALL_VALU_MODEL, "frac": adds / MIXED_ADD_ALL_VALU_MODEL,
                               "note": "all 4818 VALU instructions of the path priced by class (2.20 / 4.12 / 4.27 cycles): "
                                       "18.7 k cycles per wavefront and addition at 2400 MHz; the kernel holds 2.06-2.2 GHz "
                                       "(power-limited) and the whole-addition micro-kernel measures 19.5 k cycles"},
            "kernel_ms": t_acc * 1e3, "launches_per_msm": launches, "terms_per_launch": terms_per_launch,
            "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "algorithmic_bytes": alg_bytes, "traffic": traffic,
                    "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, "
                                      "bytes per launch)"},
            "hbm_frac": ach / HBM_PEAK_GBS,
            "sq_valu_util": _sq("msm", "ty::msm_accum_kernel grid=524288").get("valu_util"),
            "effective_clock_ghz": _sq("msm", "ty::msm_accum_kernel grid=524288").get("effective_clock_ghz"),
            "note": "integer-VALU-bound (92 % of the issue slots at an effective 2.2 GHz, profiles/r0x_pmc_sq_valu_msm.json); the HBM "
                    "fraction the north-star asks for is kept as hbm_frac; PMC traffic is "
                    + (f"{traffic / alg_bytes:.1f}" if traffic else "~19.5") + " x the algorithmic bytes (13 window gathers of a "
                    "128-B-stride record each + bucket store / reload, served by the Infinity Cache) -- see DESIGN.md section 4"}

    if rank == 0 and not args.msm_only:
        with Section(result, "ntt"):
            bench_ntt(ctx, n, log_n, device, result)
    if rank == 0 and world == 1 and not args.msm_only:
        with Section(result, "prove_hotpath"):
            bench_prove_sequence(ctx, sh, n, log_n, device, result)
        with Section(result, "prove"):
            bench_prove(ctx, sh, log_n, result)
        if log_n == 20 and not args.no_prove_2_22:
            with Section(result, "prove_2_22"):
                bench_prove_2_22(dev_index, result)
        if not args.no_cpu_baseline:
            with Section(result, "cpu_baseline"):
                ok = bench_cpu(args, ctx, sh, full, n, log_n, secret, out_xy, out_inf, result)
                if not ok:
                    result["value"] = None
                    result["error"] = "GPU result differs from the oracle: number withheld"
            if "cpu_baseline_error" in result:   # the parity gate did not run: the number is unchecked
                result["value"] = None

    if world > 1 and not args.no_sharded_prove and not args.msm_only:
        bench_sharded_prove(ctx, sh, log_n, world, backend, device, result)
    if world > 1 and rank == 0:
        if not args.msm_only:
            with Section(result, "one_gpu_same_run"):
                bench_one_gpu_same_run(dev_index, log_n, device, result)
        add_scaling_context(result, log_n, world)

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


def bench_ntt(ctx, n, log_n, device, result) -> None:
    """NTT 2^log_n, resident data"""
    v = synthetic_scalars(n, 0xA11CE, device)
    for _ in range(2):
        ctx.ntt_devptr(v.data_ptr(), log_n)
    torch.cuda.synchronize()
    # forward and inverse transforms alternate (a round trip leaves the data unchanged); both run on the 9 x 30-bit kernel
    # since round 4 (ntt_host.hip ntt_run), the inverse one with its n^-1 factor folded into the last pass
    reps, kdir = 10, [0.0, 0.0]
    t1 = time.perf_counter()
    for i in range(reps):
        ctx.ntt_devptr(v.data_ptr(), log_n, inverse=bool(i & 1))
        kdir[i & 1] += prof_ms(ctx, "ntt_")
    torch.cuda.synchronize()
    tn = (time.perf_counter() - t1) / reps
    kern = kdir[0] + kdir[1]
    kern_s = kdir[0] / (reps // 2) * 1e-3     # the VALU figures below describe the forward kernel (profiles/r03_isa_ntt.json)
    # Fr multiplications the kernels execute per transform: n/2 log2(n) butterflies less the
    # twiddle-1 ones of the last two stages of every pass, plus one inter-pass / scaling factor per element and pass
    fr_muls = _isa("r04_isa_ntt30.json", f"fr_mul_per_transform_2_{log_n}", n * (log_n / 2.0 + 1))
    mean_s = kern / reps * 1e-3
    result["ntt"] = {"log_n": log_n, "ms": tn * 1e3, "kernel_ms": kern / reps,
                     "forward_kernel_ms": kdir[0] / (reps // 2), "inverse_kernel_ms": kdir[1] / (reps // 2),
                     "algorithmic_GBps": 64.0 * n / mean_s / 1e9, "frac_of_hbm_peak": 64.0 * n / mean_s / 1e9 / HBM_PEAK_GBS,
                     "valu": {"achieved": fr_muls / kern_s, "peak": FR_MUL_MULTIPLIER_CEILING, "unit": "Fr mul/s",
                              "frac": fr_muls / kern_s / FR_MUL_MULTIPLIER_CEILING,
                              "fr_mul_per_transform": fr_muls,
                              "peak_source": "153.75 v_mad_u64_u32 per Fr multiplication on nine 30-bit limbs (profiles/r04_isa_ntt30.json) "
                                             "x 4.27 cycles at 2400 MHz x 1024 SIMDs x 64 lanes (multiplier instructions only)",
                              "all_valu_model": {"peak": FR_MUL_ALL_VALU_MODEL, "frac": fr_muls / kern_s / FR_MUL_ALL_VALU_MODEL,
                                                 "note": "all 1155 VALU instructions of a radix-4 group (4 multiplications, 8 lazy "
                                                         "additions / subtractions, addressing) priced by class: 4156 cycles"},
                              "sq_valu_util": _sq("ntt", "ty::ntt_pass30_kernel grid=%d" % (n // 4)).get("valu_util")}}
    # the same transform in a GROUP of three (typlonk_ntt_fr_batch_devptr: every pass one launch carrying three vectors' tiles)
    # -- how the reference issues its interpolations (proof.rs:50, 113-115, 334-338) and how round 1 of typlonk_prove does
    try:
        cnt = 3
        vb = synthetic_scalars(n * cnt, 0xB47C4, device)
        ptrs = [vb.data_ptr() + 32 * n * k for k in range(cnt)]
        for _ in range(2):
            ctx.ntt_batch_devptr(ptrs, log_n)
        torch.cuda.synchronize()
        kb = 0.0
        t1 = time.perf_counter()
        for i in range(reps):
            ctx.ntt_batch_devptr(ptrs, log_n, inverse=bool(i & 1))
            kb += prof_ms(ctx, "ntt_")
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t1) / reps
        result["ntt"]["batched"] = {"count": cnt, "kernel_ms_per_transform": kb / reps / cnt, "ms_per_transform": tb / cnt * 1e3,
                                    "algorithmic_GBps": 64.0 * n * cnt / (kb / reps * 1e-3) / 1e9,
                                    "note": "typlonk_ntt_fr_batch_devptr, forward and inverse alternating; bit-identical to single calls "
                                            "(tests/test_gpu_ntt.py)"}
    except Exception as e:  # noqa: BLE001 -- a secondary figure
        result["ntt"]["batched_error"] = f"{type(e).__name__}: {e}"
    # PMC traffic of one pass (the 30-bit kernel since round 4; the older files describe the 8 x 32 kernel)
    for name, key in (("r06_pmc_ntt.json", f"ntt_pass30_kernel n=2^{log_n}"), ("r05_pmc_ntt.json", f"ntt_pass30_kernel n=2^{log_n}"),
                      ("r03_pmc_ntt.json", f"ntt_pass_kernel n=2^{log_n}")):
        pm = _isa(name, key, None)
        if pm:
            result["ntt"]["pmc_traffic_bytes_per_pass"] = pm["traffic_bytes_per_pass"]
            other = _isa(name, key + " wg1024", None)   # the two passes of the 2^20 transform run different workgroup sizes
            if other:
                result["ntt"]["pmc_traffic_bytes_per_transform"] = pm["traffic_bytes_per_pass"] + other["traffic_bytes_per_pass"]
            result["ntt"]["pmc_source"] = f"profiles/{name} ({key})"
            break
    result["ntt"]["note"] = ("VALU-bound arithmetic with exposed memory phases at this size (DESIGN.md section 5): the algorithmic "
                             "figure counts 64 B per element once; the passes of a transform move more "
                             "(pmc_traffic_bytes_per_pass x passes: data in, data out and a full twiddle table)")


def bench_prove_sequence(ctx, sh, n, log_n, device, result) -> None:
    """kernel sequence of one prove(): 15 size-n NTTs, the quotient on the 4n coset domain
    (per-circuit constants cached by typlonk_circuit_load), 13 MSMs in the groups prove() issues them.  Timing only:
    the polynomials are random, not a satisfying witness."""
    bufs = [ctx.alloc(n) for _ in range(13)]
    for i, bf in enumerate(bufs):
        bf.upload(synthetic_scalars(n, 0xB0B + i, device).cpu().numpy().view(np.uint64))
    wires, zbuf, pibuf, sel, sig = bufs[0:3], bufs[3], bufs[4], bufs[5:10], bufs[10:13]
    t_out = ctx.alloc(4 * n)
    cid = ctx.circuit_load(log_n, sel, sig)
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

    try:
        for batched, key in ((False, "prove_hotpath_sequential_ms"), (True, "prove_hotpath_ms")):
            prove_sequence(batched)  # warm-up (allocates workspaces / tables on first use)
            t1 = time.perf_counter()
            tq = prove_sequence(batched)
            result[key] = (time.perf_counter() - t1) * 1e3
            result["quotient_ms"] = tq
    finally:
        ctx.circuit_free(cid)
        for bf in bufs + [t_out]:
            bf.free()


def bench_prove(ctx, sh, log_n, result) -> None:
    """a real prove(): squaring-chain circuit with n = 2^log_n rows, satisfying witness, the whole device-side flow of
    plonk::proof::prove (rounds 1-3; Fiat-Shamir challenges injected)"""
    from typlonk_amd.circuits import SquaringChain

    chain = SquaringChain(ctx, log_n)
    try:
        ch = [fr_mont_limbs(0x1234567 + k) for k in range(4)]
        zero_limbs = np.zeros(4, dtype=np.uint64)

        def run_prove():
            return ctx.prove(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,
                             lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))

        # (building the circuit on the host has let the device idle: proofs for half a second bring its clocks back up before
        # the timed ones, as in the CPU section below -- without this the same proof reads 1-2 ms more)
        def timed(fn, reps=5, warm_s=0.5):
            t0 = time.perf_counter()
            out = fn()
            while time.perf_counter() - t0 < warm_s:
                out = fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps * 1e3, out

        result["prove_ms"], proof = timed(run_prove)
        result["prove_valid"] = bool((proof["evals"][5] == zero_limbs).all())   # r(zeta) == 0, proof.rs:234-235
        result["prove_config"] = f"squaring chain, {chain.gates} gates, n = 2^{log_n}, 7 commitments + 6 openings"
        # same proof with the openings at zeta batched (typlonk_prover_round3_evals / round4_batched): 9 MSMs
        run_b = lambda: ctx.prove(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,   # noqa: E731
                                  lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]), challenge_v=lambda e: ch[1])
        result["prove_batched_openings_ms"], _ = timed(run_b, warm_s=0.0)
        # the same proof through the one-call native entry point (typlonk_prove: the reference's own Fiat-Shamir
        # transcript, restated natively, between the rounds -- no Python inside the proof)
        result["prove_native_ms"], pn = timed(lambda: ctx.prove_native(sh.sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets),
                                              warm_s=0.0)
        result["prove_native_valid"] = bool((pn["evals"][5] == zero_limbs).all())
    finally:
        chain.free()


def bench_prove_2_22(dev_index, result) -> None:
    """BASELINE config 5's size on this GPU, next to the 2^20 headline (review of round 4): the 2^22-row squaring chain,
    quotient on the 2^24 coset, in the reference's proof shape through the one-call native prover and with batched
    openings; its own context and SRS (tables c = 20), released before the CPU legs.  r(zeta) = 0 is checked here; the
    proofs at this size are compared element for element with the single-rank / eight-rank / CPU provers by the test suite
    (tests/test_gpu_prove.py, tests/test_gpu_dist.py) and by `bench.py --log-n 22`."""
    from typlonk_amd.circuits import SquaringChain

    log_n = 22
    c2 = typlonk_amd.Context(dev_index)
    try:
        sid = c2.srs_generate(fr_mont_limbs(2), (1 << log_n) + 3)
        c2.srs_precompute(sid, 20)
        chain = SquaringChain(c2, log_n)
        ch = [fr_mont_limbs(0x1234567 + k) for k in range(4)]
        zero = np.zeros(4, dtype=np.uint64)
        out = {"log_n": log_n}
        pn = c2.prove_native(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            pn = c2.prove_native(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
        torch.cuda.synchronize()
        out["prove_native_ms"] = (time.perf_counter() - t1) / 2 * 1e3
        out["valid"] = bool((pn["evals"][5] == zero).all())
        run_b = lambda: c2.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,   # noqa: E731
                                 lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]), challenge_v=lambda e: ch[1])
        run_b()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            run_b()
        torch.cuda.synchronize()
        out["prove_batched_openings_ms"] = (time.perf_counter() - t1) / 2 * 1e3
        out["config"] = f"squaring chain, {chain.gates} gates, n = 2^{log_n}, quotient on the 2^{log_n + 2} coset"
        result["prove_2_22"] = out
        chain.free()
    finally:
        c2.close()


def bench_cpu(args, ctx, sh, full, n, log_n, secret, out_xy, out_inf, result) -> bool:
    """parity gate + the CPU baselines of BASELINE.md section 3 (the oracle is the checker, timed on bounded samples):
    R1/R2 reference-faithful MSM (1 core, as the reference), R3 schoolbook quotient (extrapolated), F1 bucket-method MSM
    on all cores at full size, F2 all-core NTT and a fair CPU prove() next to the GPU prove() of the same circuit."""
    from oracle import coracle as CO

    ms = min(args.cpu_sample, n)
    sc_all = full.cpu().numpy().view(np.uint64)
    sc_host = sc_all[:ms]
    xy, inf = ctx.srs_download(sh.sid, 0, ms)
    CO.group_ops_reset()
    t1 = time.perf_counter()
    ref_xy, ref_inf = CO.msm_reference(sc_host, xy, inf)
    tc = time.perf_counter() - t1
    cpu_ops = CO.group_ops()
    gxy, ginf = ctx.msm_devptr(sh.sid, full.data_ptr(), ms)
    parity_sample = bool((gxy == ref_xy).all() and ginf == ref_inf)
    # full size: commit(p) == [p(s)]G  (the reference's own test identity, kzg/src/lib.rs:102-105)
    ps = CO.poly_eval(sc_all, secret)
    exp_xy, exp_inf = CO.g1_mul_generator(ps)
    parity_full = bool((np.asarray(out_xy) == exp_xy).all() and out_inf == exp_inf)
    result["parity"] = {"sample_vs_oracle": parity_sample, "full_commit_identity": parity_full}
    result["cpu_baseline"] = {
        "value": cpu_ops / tc, "unit": "G1-adds/s", "cores": 1, "kind": "port",
        "sample": f"first {ms} terms of the same scalar/SRS vectors, reference-faithful per-term "
                  f"double-and-add + affine normalisation + sum (kzg/src/lib.rs:41-54)",
        "terms_per_s": ms / tc, "seconds": tc, "host_cpus": os.cpu_count()}
    # F1 "fair CPU" (SURVEY 8d-ii): bucket method on all host cores (oracle_msm_pippenger) on the FULL vector
    xyf, inff = ctx.srs_download(sh.sid, 0, n)
    cf = max(8, min(13, log_n - 5))   # swept on the 2 x 64-core host (tools/cpu_msm_sweep.py): buckets that fit the L2 win
    CO.msm_pippenger(sc_all[:1024], xyf[:1024], inff[:1024], c=8)      # spin the OpenMP team up
    t1 = time.perf_counter()
    f_xy, f_inf, f_ops, f_thr = CO.msm_pippenger(sc_all, xyf, inff, c=cf)
    tf = time.perf_counter() - t1
    parity_fair = bool((np.asarray(out_xy) == f_xy).all() and out_inf == f_inf)
    result["parity"]["full_vs_fair_cpu"] = parity_fair
    result["cpu_fair"] = {
        "kind": "port (bucket method, not the reference's algorithm)", "cores": f_thr, "window_bits": cf,
        "sample": f"all {n} terms", "msm_seconds": tf, "terms_per_s": n / tf, "value": f_ops / tf, "unit": "G1-adds/s"}
    # F2: radix-2 NTT at full size, one thread (as the reference) and all cores
    for threads, key in ((1, "ntt_ms_1_thread"), (0, "ntt_ms_all_cores")):
        buf = sc_all.copy()
        t1 = time.perf_counter()
        CO.ntt(buf, log_n, threads=threads)
        result["cpu_fair"][key] = (time.perf_counter() - t1) * 1e3
    result["cpu_fair"]["ntt_log_n"] = log_n
    # R3: the reference's schoolbook quotient (12 naive_mul, plonk/src/proof.rs:317-359), measured small, extrapolated ~ n^2
    # (--cpu-full: the three octaves SURVEY 8d asks for, 2^10 / 2^12 / 2^14 -- the last one ~3 min on one core; the default
    # run measures 2^10 and 2^11 and leans on profiles/r06_bench_cpu_full.json for the law)
    sizes = [10, 11] + ([12, 14] if args.cpu_full else [])
    quad = {}
    for lg in sizes:
        t1 = time.perf_counter()
        CO.quotient_schoolbook_products(sc_all[: 1 << lg], 1 << lg)
        quad[lg] = time.perf_counter() - t1
    lg = sizes[-1]
    import math

    fit = math.log(quad[sizes[-1]] / quad[sizes[0]], 2.0) / (sizes[-1] - sizes[0])   # measured exponent of n (2.0 = the n^2 law)
    result["cpu_reference_quotient"] = {
        "kind": "port, 1 core", "measured_s": {f"2^{k}": v for k, v in quad.items()},
        "measured_exponent_of_n": fit,
        f"extrapolated_s_2^{log_n}": quad[lg] * 4.0 ** (log_n - lg),
        "full_measurement": "profiles/r06_bench_cpu_full.json (bench.py --cpu-full: 2^10, 2^11, 2^12 and 2^14 measured on one core)",
        "note": f"the 12 schoolbook products of quotient_polynomial (19 n^2 multiply-adds), extrapolated from 2^{lg} with "
                "the n^2 law -- the reference's own prove() is dominated by this term"}
    if args.cpu_full:   # R2: the reference's MSM on the full vector
        CO.group_ops_reset()
        t1 = time.perf_counter()
        r_xy, r_inf = CO.msm_reference(sc_all, xyf, inff)
        tr = time.perf_counter() - t1
        result["cpu_baseline_full"] = {"terms": n, "seconds": tr, "terms_per_s": n / tr, "value": CO.group_ops() / tr,
                                       "unit": "G1-adds/s", "cores": 1,
                                       "equals_gpu": bool((np.asarray(out_xy) == r_xy).all() and out_inf == r_inf)}
    # F2: a fair CPU prove() (all cores: NTT quotient, batch-inverted grand product, bucket-method MSMs) on the
    # squaring chain at 2^16, end to end, next to the GPU prove() of the SAME circuit, witness and challenges
    with Section(result, "cpu_fair_prove"):
        from oracle import cpu_prover as CP
        from typlonk_amd.circuits import SquaringChain

        # ... and at the headline size itself (one CPU proof, ~25 s on 128 cores): the measured denominator of the
        # north-star's ">= 10x prove()" -- no extrapolation
        # (on a small host the full-size CPU proof would take minutes: only with >= 32 cores, or with --cpu-full)
        full_size_too = (os.cpu_count() or 1) >= 32 or args.cpu_full
        for lg in sorted({min(16, log_n), log_n} if full_size_too else {min(16, log_n)}):
            chain = SquaringChain(ctx, lg, keep_host=True)
            try:
                own_srs = lg != log_n or sh.world != 1     # the headline SRS (c = 20 tables) serves the headline size
                sid_l = sh.sid
                if own_srs:
                    sid_l = ctx.srs_generate(secret, (1 << lg) + 3)
                    ctx.srs_precompute(sid_l, 0)   # the library's choice by length (c = 15 below 2^16 points, 17 below 2^19)
                ch = [fr_mont_limbs(0x1234567 + k) for k in range(4)]
                run = lambda: ctx.prove(sid_l, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,   # noqa: E731,B023
                                        lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))                   # noqa: B023
                # the GPU has idled through the CPU legs above: proofs for half a second bring its clocks back up before
                # the timed ones (without this the same proof reads 51 ms instead of 37 at 2^20)
                t1 = time.perf_counter()
                while time.perf_counter() - t1 < 0.5:
                    run()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    gp = run()
                torch.cuda.synchronize()
                t_gpu = (time.perf_counter() - t1) / 3 * 1e3
                inputs = chain.host_inputs()
                sxy, sinf = ctx.srs_download(sid_l, 0, (1 << lg) + 3)
                if lg <= 16:
                    CP.prove(lg, inputs, sxy, sinf, ch)                   # warm-up (OpenMP team, page faults)
                t1 = time.perf_counter()
                cp = CP.prove(lg, inputs, sxy, sinf, ch)
                t_cpu = (time.perf_counter() - t1) * 1e3
                same = all((np.asarray(a[0]) == np.asarray(b[0])).all() and a[1] == b[1]
                           for k in ("commit", "t_commit", "witness") for a, b in zip(gp[k], cp[k]))
                same = same and bool((gp["z_commit"][0] == cp["z_commit"][0]).all())
                same = same and all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(gp["evals"], cp["evals"]))
                result["cpu_fair"].update({f"prove_ms_2_{lg}": t_cpu, f"gpu_prove_ms_2_{lg}": t_gpu,
                                           f"prove_speedup_2_{lg}": t_cpu / t_gpu, f"prove_stage_ms_2_{lg}": cp["stage_ms"],
                                           f"prove_equals_gpu_2_{lg}": bool(same), "prove_threads": cp["threads"]})
                result["parity"]["prove_vs_fair_cpu_prover"] = bool(same) and result["parity"].get("prove_vs_fair_cpu_prover", True)
                if own_srs:
                    ctx.srs_free(sid_l)
            finally:
                chain.free()
    # the north-star's ">= 10x prove() over the CPU reference"
    if "prove_ms" in result:
        msm_terms = 13 * n
        fair_prove_s, fair_note = None, "not measured"
        if f"prove_ms_2_{log_n}" in result["cpu_fair"]:
            fair_prove_s = result["cpu_fair"][f"prove_ms_2_{log_n}"] * 1e-3
            fair_note = (f"measured end to end at 2^{log_n} on {result['cpu_fair'].get('prove_threads')} threads, same circuit, "
                         "witness and challenges as the GPU proof, every proof element equal")
        elif f"prove_ms_2_{min(16, log_n)}" in result["cpu_fair"]:
            # a prove() is ~linear in n on the CPU (bucket MSMs and NTTs dominate): scale the measured 2^16 proof
            fair_prove_s = result["cpu_fair"][f"prove_ms_2_{min(16, log_n)}"] * 1e-3 * (n / (1 << min(16, log_n)))
            fair_note = (f"measured end to end at 2^{min(16, log_n)} on {result['cpu_fair'].get('prove_threads')} threads, "
                         f"scaled linearly to 2^{log_n}")
        result["prove_vs_cpu"] = {
            "gpu_prove_ms": result["prove_ms"],
            "cpu_reference_path_s": msm_terms / (ms / tc) + result["cpu_reference_quotient"][f"extrapolated_s_2^{log_n}"],
            "cpu_reference_path_parts": "13 MSMs at the measured reference-path rate + the extrapolated schoolbook quotient",
            "cpu_reference_cores": 1,
            "cpu_fair_prove_s": fair_prove_s,
            "cpu_fair_note": fair_note,
            "cpu_fair_13_msms_s": 13 * tf, "cpu_all_cores": f_thr,
            "speedup_vs_reference_path": (msm_terms / (ms / tc) + result["cpu_reference_quotient"][f"extrapolated_s_2^{log_n}"])
                                         / (result["prove_ms"] * 1e-3),
            "speedup_vs_fair_cpu": (fair_prove_s / (result["prove_ms"] * 1e-3)) if fair_prove_s else None,
            "speedup_vs_fair_cpu_msms_only": 13 * tf / (result["prove_ms"] * 1e-3)}
    ok = parity_sample and parity_full and parity_fair and result["parity"].get("prove_vs_fair_cpu_prover", True)
    return bool(ok)


def bench_sharded_prove(ctx, sh, log_n, world, backend, device, result) -> None:
    """prove() with every MSM sharded over the ranks (NTT / quotient replicated): typlonk_amd.dist.ShardedProver.
    The block contains collectives, so the ranks AGREE on success before entering it; a failure inside the timed proofs
    is recorded as prove_sharded_error (the peers' collectives then end with the process-group timeout)."""
    from typlonk_amd.circuits import SquaringChain
    from typlonk_amd.dist import ShardedProver

    def agree(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def sync_all():
        dist.barrier()
        torch.cuda.synchronize()

    chain, setup_err = None, None
    try:
        chain = SquaringChain(ctx, log_n)
    except Exception as e:  # noqa: BLE001 -- setup only: no collective has been entered yet
        setup_err = f"{type(e).__name__}: {e}"
    try:
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
            # the reference's proof shape (six openings, 13 MSMs): the figure to hold against the 1-GPU prove_ms
            run_6 = lambda: sp.prove(chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,   # noqa: E731
                                     lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))
            run_6()
            sync_all()
            t1 = time.perf_counter()
            for _ in range(3):
                run_6()
            sync_all()
            result["prove_sharded_ms"] = (time.perf_counter() - t1) / 3 * 1e3
            result["prove_valid"] = bool((proof["evals"][5] == np.zeros(4, dtype=np.uint64)).all())
            if sh.native:   # the one-call native prover on the shard: the library folds every round itself
                sp.prove_native(chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
                sync_all()
                t1 = time.perf_counter()
                for _ in range(3):
                    pn = sp.prove_native(chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets)
                sync_all()
                result["prove_sharded_native_ms"] = (time.perf_counter() - t1) / 3 * 1e3
                result["prove_native_valid"] = bool((pn["evals"][5] == np.zeros(4, dtype=np.uint64)).all())
        else:
            result["prove_sharded_error"] = setup_err or "setup failed on another rank"
    except Exception as e:  # noqa: BLE001
        result["prove_sharded_error"] = f"{type(e).__name__}: {e}"
        traceback.print_exc(file=sys.stderr)
    finally:
        if chain is not None:
            chain.free()


if __name__ == "__main__":
    main()
