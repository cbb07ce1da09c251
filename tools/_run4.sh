export TMPDIR=/tmp
O=gpurun_out/r03_d; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_msm_shard.py -x -q) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
(timeout 600 python bench.py --steps 20 --warmup 5) > $O/bench.out 2> $O/bench.err; tail -1 $O/bench.out > $O/bench.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_d/bench.json"))
for k in ("value","ms_per_step","msm_stage_ms","ntt","prove_ms","prove_batched_openings_ms","prove_native_ms","parity","cpu_fair","cpu_reference_quotient","prove_vs_cpu"):
    print(k, d.get(k))
print({k:v for k,v in d.items() if k.endswith("error")})
print(d["roofline"]["frac"], d["roofline"]["hbm_frac"], d["roofline"]["all_valu_model"])
PY
tail -5 $O/bench.err
