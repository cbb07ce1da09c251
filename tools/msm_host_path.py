"""typlonk_msm_g1 with the scalars in HOST memory (what the reference's commit() hands over) against the device-resident form:
wall time per call at 2^20 and 2^22 terms -- the PCIe-inclusive rate DESIGN.md section 8 quotes next to `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

ctx = typlonk_amd.Context(0)
for log_m in [int(x) for x in os.environ.get("SIZES", "20,22").split(",")]:
    m = 1 << log_m
    sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
    host = sc.cpu().numpy().view(np.uint64).reshape(m, 4).copy()
    sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
    ctx.srs_precompute(sid, 20)
    ref = ctx.msm_devptr(sid, sc.data_ptr(), m)
    for _ in range(3):
        got = ctx.msm(sid, host)
    assert (np.asarray(got[0]) == np.asarray(ref[0])).all() and got[1] == ref[1]
    reps = 20 if log_m <= 20 else 6
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm(sid, host)
    th = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_devptr(sid, sc.data_ptr(), m)
    td = (time.perf_counter() - t0) / reps * 1e3
    print(f"HOSTPATH 2^{log_m}: host scalars {th:.3f} ms per MSM, device-resident {td:.3f} ms, difference {th - td:.3f} ms "
          f"({32 * m / 1e6:.0f} MB over PCIe)", flush=True)
    ctx.srs_free(sid)
