"""Two independent 2^LOG_N transforms on two contexts / streams against the same two one after the other: how much of a
stand-alone transform's time is memory phase that another transform's arithmetic can hide."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from bench import synthetic_scalars

dev = torch.device("cuda", 0)
for log_n in [int(x) for x in os.environ.get("SIZES", "16,18,20,22").split(",")]:
    n = 1 << log_n
    ctxs = [typlonk_amd.Context(0) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for c, s in zip(ctxs, streams):
        c.set_stream(s.cuda_stream)
    xs = [synthetic_scalars(n, 3 + i, dev) for i in range(2)]
    def run(pairs, reps=50):
        for _ in range(5):
            for c, x in pairs: c.ntt_devptr(x.data_ptr(), log_n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for c, x in pairs: c.ntt_devptr(x.data_ptr(), log_n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    one = run([(ctxs[0], xs[0])])
    serial = run([(ctxs[0], xs[0]), (ctxs[0], xs[1])])
    conc = run([(ctxs[0], xs[0]), (ctxs[1], xs[1])])
    print(json.dumps({"log_n": log_n, "one_ms": round(one, 4), "two_on_one_stream_ms": round(serial, 4), "two_on_two_streams_ms": round(conc, 4)}), flush=True)
    for c in ctxs: c.close()
