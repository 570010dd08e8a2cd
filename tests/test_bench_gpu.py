"""bench.py's contract line, run as the driver runs it (a subprocess): N = 1, and the N > 1 code path rehearsed on
one GPU with a one-rank RCCL group (the per-step loss all-reduce captured into the hipGraph)."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline", "--no-eager-python", *args],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_n1():
    d = _bench()
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["metric"] == "ctc_samples_per_sec" and d["unit"] == "samples/s" and d["vs_baseline"] is None
    assert math.isfinite(d["value"]) and d["value"] > 1e6
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] < 1.0
    assert d["parity"]["max_abs_err_grad"] < 1e-6
    assert d["config"]["collective"] is None


def test_bench_collective_path_rehearsal():
    d = _bench("--rehearse-collective")
    c = d["config"]
    assert c["collective"] == "nccl" and c["loss_allreduce_bucket"] == 1 and c["launch"] == "graph"
    assert c["collective_launch"].startswith("in the hipGraph")
    assert math.isfinite(d["value"]) and d["value"] > 1e6
    e = _bench("--rehearse-collective", "--collective-launch", "eager")
    assert e["config"]["launch"] == "eager" and e["config"]["collective_launch"].startswith("eager")
    assert math.isfinite(e["value"]) and e["value"] > 1e5
