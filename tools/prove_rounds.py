"""Wall-clock per prover round (typlonk_prover_round1/2/3/4) for the squaring-chain circuit."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from typlonk_amd.capi import ProofTail, ProofEvals, ProofBatched, _u64p, _u8p
from typlonk_amd.circuits import SquaringChain
from bench import fr_mont_limbs

log_n = int(os.environ.get("LOG_N", "20"))
ctx = typlonk_amd.Context(0)
n = 1 << log_n
sid = ctx.srs_generate(fr_mont_limbs(2), n + 3)
if int(os.environ.get("TABLES", "20")):
    ctx.srs_precompute(sid, int(os.environ.get("TABLES", "20")))
chain = SquaringChain(ctx, log_n)
ch = [fr_mont_limbs(0x1234567 + k) for k in range(5)]
lib = ctx.lib
for batched in (False, True):
    for rep in range(3):
        w = (C.c_void_p * 3)(*[b.handle.value for b in chain.wire_evals])
        pr = C.c_void_p()
        cxy, cinf = ((C.c_uint64 * 12) * 3)(), (C.c_uint8 * 3)()
        ks = ((C.c_uint64 * 4) * 3)()
        for i in range(3):
            for j, limb in enumerate(np.asarray(chain.cosets[i], dtype=np.uint64).reshape(4)):
                ks[i][j] = int(limb)
        zxy, zinf = np.zeros(12, dtype=np.uint64), np.zeros(1, dtype=np.uint8)
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        ctx._chk(lib.typlonk_prover_round1(ctx.h, sid, chain.circuit, w, None, C.byref(pr), C.byref(cxy), C.byref(cinf)))
        t.append(time.perf_counter())
        ctx._chk(lib.typlonk_prover_round2(pr, _u64p(ch[0]), _u64p(ch[1]), C.byref(ks), _u64p(zxy), _u8p(zinf)))
        t.append(time.perf_counter())
        if batched:
            pe = ProofEvals()
            ctx._chk(lib.typlonk_prover_round3_evals(pr, _u64p(ch[2]), _u64p(ch[3]), C.byref(pe)))
            t.append(time.perf_counter())
            pb = ProofBatched()
            ctx._chk(lib.typlonk_prover_round4_batched(pr, _u64p(ch[4]), C.byref(pb)))
        else:
            tail = ProofTail()
            ctx._chk(lib.typlonk_prover_round3(pr, _u64p(ch[2]), _u64p(ch[3]), C.byref(tail)))
        t.append(time.perf_counter())
        lib.typlonk_prover_free(pr)
    print("batched" if batched else "six-opening", [round((b - a) * 1e3, 2) for a, b in zip(t, t[1:])], "ms; total",
          round((t[-1] - t[0]) * 1e3, 2))
