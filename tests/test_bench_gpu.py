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
    """`--rehearse-collective` runs exactly what an N > 1 run does by default: one all-reduce of the scalar per step
    (the north_star form), captured into the hipGraph on the launch stream, the bucketed form as the secondary figure."""
    d = _bench("--rehearse-collective")
    c = d["config"]
    assert c["collective"] == "nccl" and c["loss_allreduce_bucket"] == 1 and c["launch"] == "graph"
    assert c["collective_launch"].startswith("in the hipGraph") and "launch stream" in c["collective_launch"]
    assert "loss all-reduce every step" in c["workload"] and "note" not in c, c
    assert math.isfinite(d["value"]) and d["value"] > 1e6
    assert d["bucketed"]["loss_allreduce_bucket"] == 20 and math.isfinite(d["bucketed"]["value"])
    b = _bench("--rehearse-collective", "--loss-bucket", "20")
    assert b["config"]["loss_allreduce_bucket"] == 20 and "every 20 steps" in b["config"]["workload"]
    assert math.isfinite(b["per_step_collective"]["value"])
    e = _bench("--rehearse-collective", "--collective-launch", "eager")
    assert e["config"]["launch"] == "eager" and e["config"]["collective_launch"].startswith("eager")
    assert math.isfinite(e["value"]) and e["value"] > 1e5


def test_bench_collective_coresidence_measurements():
    """A one-rank all-reduce launches no kernel; `--occupant K` puts workgroups with RCCL's kernel footprint where each
    step's all-reduce kernel would run on a side stream (a kernel node on a second stream is what
    costs a captured graph its time) and also times the same-stream placement the default uses."""
    d = _bench("--rehearse-collective", "--occupant", "2")
    co = d["coresident"]
    for k in ("gated_us_per_step", "ungated_us_per_step", "allreduce_only_us_per_step", "allreduce_same_stream_us_per_step",
              "no_collective_us_per_step"):
        assert co[k] is not None and math.isfinite(co[k]) and co[k] > 5.0, co
    assert abs(co["allreduce_only_us_per_step"] - co["no_collective_us_per_step"]) < 2.0, co
    assert abs(co["allreduce_same_stream_us_per_step"] - co["no_collective_us_per_step"]) < 2.0, co
    assert co["ungated_us_per_step"] > co["no_collective_us_per_step"] + 3.0, co


def test_collective_gate_returns_and_arrival_words_reset():
    """ctc_amd_collective_gate: the first gate on a workspace switches the counting of arrivals on; a later gate
    opens when the next loss launch has filled the chip and runs into its bound when none follows; the launch's last
    workgroup puts the sixteen arrival words back to 0."""
    import time
    import torch
    import ctc_amd
    from ctc_amd import functional as F
    from tests.helpers import synth_noblank
    dev = torch.device("cuda:0")
    x, lab, Tb, L = synth_noblank(0, 150, 256, 158, 20)
    args = (x.to(dev).requires_grad_(True), lab.to(dev), Tb.to(dev), L.to(dev))
    assert F.collective_gate("noblank", 256, dev, launch_stream=123456789) is False      # no such stream's workspace
    ctc_amd.CTCLoss.apply(*args)
    torch.cuda.synchronize()
    key = (dev.index, F._stream_handle(dev), 0)
    ws = F._workspaces[key][-1]
    words = ws[:512].view(torch.int32)

    def arrivals():
        return int(sum(int(words[64 + 4 * sh + 2]) for sh in range(16)))
    assert int(words[11]) == 0 and arrivals() == 0                                      # counting is off until a gate is used
    side = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        assert F.collective_gate("noblank", 256, dev, launch_stream=main.cuda_stream, timeout_us=10)   # switches it on
    torch.cuda.synchronize()
    assert int(words[11]) == 1
    for follow in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            assert F.collective_gate("noblank", 256, dev, launch_stream=main.cuda_stream, timeout_us=20000)
        if follow:
            ctc_amd.CTCLoss.apply(*args)                                                 # fills the chip: the gate opens
        side.synchronize()
        el = time.perf_counter() - t0
        assert (el < 0.015) if follow else (0.015 < el < 0.2), (follow, el)             # opened by the launch / by its bound
        torch.cuda.synchronize()
        assert arrivals() == 0                                                          # reset by the last workgroup
    assert ctc_amd.workspace_status() == 0


def test_resident_collective_kernel_costs_a_round_unless_the_launch_came_first():
    """tools/coresident.py: workgroups with RCCL's kernel footprint cannot share a CU with a loss workgroup (280 + 288
    registers per lane > 512).  Resident BEFORE the B = #CUs launch they cost it a second round of workgroups; enqueued
    after it, or held back by ctc_amd_collective_gate until the launch has filled the chip, they cost it nothing."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "coresident.py")], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["us_per_launch_event_bracketed_median"]
    base = d["loss first"]["k=0"]
    assert d["collective first"]["k=1"] > base + 4.0, d
    second_round = min(d["collective first"][k] for k in ("k=1", "k=2", "k=8"))
    for order in ("loss first", "gated"):
        for k in ("k=1", "k=2", "k=8"):                      # (a second round costs 8 us; single medians wobble by 1-3 on a busy box)
            assert d[order][k] < second_round - 3.0, (order, k, d)
        assert min(d[order][k] for k in ("k=1", "k=2", "k=8")) < base + 2.0, (order, d)
