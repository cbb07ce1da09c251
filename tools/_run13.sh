O=gpurun_out/r03_m; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; tail -1 $O/bench.out > $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_m/bench.json"))
print(d["value"], d["ms_per_step"], d["msm_stage_ms"])
print({k:v for k,v in d.items() if k.endswith("error")}, d.get("parity"))
print(d["roofline"]["frac"], d["roofline"]["hbm_frac"], d["roofline"].get("sq_valu_util"), d["ntt"]["valu"])
print(d["prove_ms"], d["prove_batched_openings_ms"], d["prove_native_ms"], d["cpu_fair"].get("prove_ms_2_16"), d["cpu_fair"].get("gpu_prove_ms_2_16"))
PY
for w in 8 4 2; do WORLD=$w REPS=300 timeout 300 python tools/shard_latency.py 2>&1 | grep "^SHARD" | tail -1 >> $O/shard_latency.jsonl; done; cat $O/shard_latency.jsonl
