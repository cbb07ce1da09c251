O=gpurun_out/r03_o; mkdir -p $O
for t in 20 17 20 17; do python bench.py --steps 30 --warmup 8 --msm-only --tables $t 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tables',$t,round(d['ms_per_step'],4),d['value'],{k:round(v,3) for k,v in d['msm_stage_ms'].items()})"; done | tee $O/c17_2_20.txt
for t in 20 17; do TABLES=$t LOG_M=20 python - <<'PY' | tee -a $O/c17_2_20.txt
import os, sys, time
sys.path.insert(0, ".")
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs
m = 1 << 20
ctx = typlonk_amd.Context(0)
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
ctx.srs_precompute(sid, int(os.environ["TABLES"]))
bufs = [synthetic_scalars(m, i, torch.device("cuda", 0)) for i in range(9)]
for _ in range(3): ctx.msm_batch_devptr(sid, [b.data_ptr() for b in bufs], [m] * 9)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): ctx.msm_batch_devptr(sid, [b.data_ptr() for b in bufs], [m] * 9)
print("batch of 9, tables", os.environ["TABLES"], round((time.perf_counter() - t) / 45 * 1e3, 4), "ms per MSM")
PY
done
