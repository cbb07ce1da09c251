"""Batched NTT (typlonk_ntt_fr_batch_devptr) against `count` single calls: HIP-event kernel time per transform.
env: SIZES (log sizes), COUNTS, TYPLONK_NTT_BIG / TYPLONK_NTT_FR30 as the library reads them."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from bench import synthetic_scalars

ctx = typlonk_amd.Context(0)
ctx.set_profiling(True)
dev = torch.device("cuda", 0)
for log_n in [int(x) for x in os.environ.get("SIZES", "20").split(",")]:
    n = 1 << log_n
    for count in [int(x) for x in os.environ.get("COUNTS", "1,3,5").split(",")]:
        x = synthetic_scalars(n * count, 3, dev)
        ptrs = [x.data_ptr() + 32 * n * v for v in range(count)]
        for inverse in (False, True):
            out = {"log_n": log_n, "count": count, "inverse": inverse, "big": os.environ.get("TYPLONK_NTT_BIG", "1")}
            for mode in ("single", "batch"):
                def run():
                    if mode == "single":
                        ks = 0.0
                        for p in ptrs:
                            ctx.ntt_devptr(p, log_n, inverse)
                            ks += sum(v for k, v in ctx.profile() if k.startswith("ntt_"))
                        return ks
                    ctx.ntt_batch_devptr(ptrs, log_n, inverse)
                    return sum(v for k, v in ctx.profile() if k.startswith("ntt_"))
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                reps, ks = 20, 0.0
                t = time.perf_counter()
                for _ in range(reps):
                    ks += run()
                torch.cuda.synchronize()
                out[mode + "_kernel_ms_per_transform"] = round(ks / reps / count, 4)
                out[mode + "_wall_ms_per_transform"] = round((time.perf_counter() - t) / reps / count * 1e3, 4)
            print(json.dumps(out), flush=True)
