"""The sharded (N > 1) MSM path of bench.py with real kernels: two ranks share the one GPU of the test box
and exchange their partial points over gloo (RCCL refuses two ranks per device; the RCCL path itself is
exercised with one rank by TYPLONK_FORCE_COLLECTIVE).  Checks the sharded result against the reference's
identity commit(p) == [p(s)]G and the JSON contract fields."""
import json
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(env_extra, nproc, args):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def test_two_ranks_sharded_msm_on_one_gpu():
    d = _run({"TYPLONK_BENCH_BACKEND": "gloo"}, 2, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--log-n", "18"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["metric"] == "msm_g1_adds_per_s"
    assert d["parity"]["full_commit_identity"] is True and d["value"] is not None
    assert "index-sharded x2" in d["config"]["parallelism"]


def test_one_rank_through_rccl_all_gather():
    d = _run({"TYPLONK_FORCE_COLLECTIVE": "1"}, 1, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "16",
                                                    "--cpu-sample", "2048"])
    assert d["n_gpus"] == 1 and d["parity"]["sample_vs_oracle"] and d["parity"]["full_commit_identity"]
    assert d["prove_valid"] is True
    for key in ("roofline", "cpu_baseline", "ms_per_step", "higher_is_better", "dtype", "data", "config"):
        assert key in d


def test_two_ranks_sharded_prove_equals_single_rank():
    """typlonk_srs_set_shard + ShardedProver: both proof shapes, every element, plus short MSMs whose range is
    empty on one rank (tests/dist_prove_worker.py)"""
    env = dict(os.environ, LOG_N="10")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_prove_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"sharded_prove_ok"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(lines[0])["sharded_prove_ok"] is True
