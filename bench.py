#!/usr/bin/env python3
"""Throughput of the CTC loss hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--variant noblank|binary|blank]
                    [--scaling weak|strong --global-batch B] [--batch B_per_gpu]

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident input:
loss AND the full input gradient (what `loss = ctc_loss(...); loss.backward()` costs in
the reference, train.py:427,444), issued through the C ABI exactly as the autograd
Function issues it: one fused `*_loss_grad` launch plus the `scale_grad` launch of
backward (upstream gradient 1.0).  Default workload = BASELINE configs[1]:
NoBlankCTC, B=256 per GPU, T=150, C=158, S<=20, fp32.

N>1 (launched by torch.distributed.run, one rank per GPU, RCCL):
  --scaling weak   (default) every rank runs the per-GPU batch (256): global batch 256*N;
  --scaling strong BASELINE configs[3]: ONE global batch (--global-batch, default 2048), seeded
                   once and sliced -- rank r owns shard_bounds(B, r, N); N=1 runs all of it.
Either way the gradient scale is 1/B_global, there is no gradient communication, and the
loss contributions are summed by RCCL with ONE all-reduce of the 4-byte scalar per step (the
BASELINE north_star; `--loss-bucket 1`, the default), captured into the hipGraph with the step
and enqueued on the launch stream itself behind `scale_grad` (`--collective-stream same`: no
cross-stream edge in the graph, no collective kernel resident beside the next loss launch;
`side` = asynchronous on RCCL's own stream).  `--loss-bucket M` sends the losses of M steps
through one all-reduce instead; the default run reports that form (M = one graph replay) as the
secondary figure `bucketed`.  A capture that fails falls back to eager issue and says so.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {                  # variant -> (name, T, C, S, default per-GPU batch, BASELINE config)
    "noblank": ("NoBlankCTC", 150, 158, 20, 256, "configs[1]"),
    "binary": ("NoBlankBinaryCTC", 150, 158, 20, 256, "configs[2]"),
    "blank": ("blank-CTC", 2000, 1000, 100, 64, "configs[4]"),
}


def workload_name(variant, B, global_batch=None, world=1, scaling="weak"):
    name, T, C, S, Bdef, cfg = WORKLOADS[variant]
    sfx = "S=%d" % S if variant == "blank" else "S<=%d" % S
    if scaling == "strong":
        return "%s B=%d (sharded over %d GPU%s) T=%d C=%d %s fp32 (BASELINE configs[3])" % (
            name, global_batch, world, "" if world == 1 else "s", T, C, sfx)
    tag = " (BASELINE %s)" % cfg if B == Bdef else " (BASELINE %s shape, batch overridden)" % cfg
    return "%s B=%d T=%d C=%d %s fp32%s" % (name, B, T, C, sfx, tag)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--variant", default="noblank", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the BASELINE config)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = --batch per GPU; strong = --global-batch sharded over the ranks (BASELINE configs[3])")
    ap.add_argument("--global-batch", type=int, default=2048, help="--scaling strong: the global batch")
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"],
                    help="graph: steps are replayed from a captured hipGraph; eager: one ctypes call per launch")
    ap.add_argument("--graph-steps", type=int, default=50, help="steps captured per hipGraph")
    ap.add_argument("--loss-bucket", type=int, default=None,
                    help="N>1: steps per RCCL all-reduce of the loss scalars (default 1 = one all-reduce per step, the north_star form)")
    ap.add_argument("--no-eager-python", action="store_true", help="skip the secondary CTCLoss.apply + backward() timing")
    ap.add_argument("--collective-launch", default="graph", choices=["graph", "eager"],
                    help="N>1, one all-reduce per step: graph = the asynchronous all-reduce of every step is captured into "
                         "the hipGraph with the step (RCCL supports capture); eager = steps and collectives issued one by one "
                         "from Python (host-bound: ~22 us per torch.distributed call)")
    ap.add_argument("--collective-stream", default="same", choices=["same", "side"],
                    help="all-reduces captured into the hipGraph: same = a blocking all-reduce on the launch stream behind "
                         "scale_grad (serial, no cross-stream dependency); side = asynchronous, forked onto a side stream")
    ap.add_argument("--watchdog-seconds", type=int, default=900,
                    help="end the process (exit code 3) if the run has not finished by then: a collective inside a "
                         "hipGraph that never completes is invisible to torch.distributed's own watchdog")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="with --gpus 1: create a one-rank process group and run the N>1 code path (all-reduce per step)")
    ap.add_argument("--collective-gate", type=int, default=0, choices=[0, 1],
                    help="collectives in the hipGraph (--loss-bucket 1): hold each step's all-reduce back (ctc_amd_collective_gate) "
                         "until the NEXT step's loss launch has filled the chip -- RCCL's kernel cannot share a CU with a loss "
                         "workgroup.  Off by default: inside a captured graph the cross-stream dependencies cost more than the gate saves")
    ap.add_argument("--occupant", type=int, default=None,
                    help="--rehearse-collective: a stand-in for the collective's kernel (a 1-rank all-reduce launches none) -- "
                         "K workgroups with rcclGenericKernel's footprint resident for --occupant-us on a side stream, and the "
                         "co-residence measurements of DESIGN.md section 5 (`coresident` in the output)")
    ap.add_argument("--occupant-us", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 collective backend: nccl = RCCL over xGMI; gloo only to rehearse the multi-rank "
                         "path on a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU budget of the oracle baseline leg")
    return ap.parse_args()


class Workload:
    """Device-resident synthetic batch + the C-ABI call sequence of one step.  `lo:hi` = this rank's
    slice of a batch of `B_gen` samples generated from `seed` (strong scaling: the global batch)."""

    def __init__(self, variant, B_gen, B_global, dev, seed, lo=0, hi=None, name=None):
        from ctc_amd import _lib
        from tests import helpers
        hi = B_gen if hi is None else hi
        self.variant, self.B, self.dev = variant, hi - lo, dev
        _, self.T, self.C, self.S = WORKLOADS[variant][:4]
        self.name = name or workload_name(variant, self.B)
        T, B, C, S = self.T, self.B, self.C, self.S
        if variant == "noblank":
            x, tg, il, tl = helpers.synth_noblank(seed, T, B_gen, C, S)
            self.target_bytes = 4 * B * S
        elif variant == "binary":
            x, tg, il, tl = helpers.synth_binary(seed, T, B_gen, C, S)
            self.target_bytes = 4 * B * S * C
        else:
            x, tg, il, tl = helpers.synth_blank(seed, T, B_gen, C, S)
            self.target_bytes = 8 * B * S
        x, tg, il, tl = x[:, lo:hi].contiguous(), tg[lo:hi].contiguous(), il[lo:hi].contiguous(), tl[lo:hi].contiguous()
        self.host = (x, tg, il, tl)
        self.x, self.tg, self.il, self.tl = (t.to(dev) for t in (x, tg, il, tl))
        self.lib = _lib.load()
        self.vid = {"noblank": _lib.NOBLANK, "binary": _lib.BINARY, "blank": _lib.BLANK}[variant]
        self.ws_bytes = self.lib.ctc_amd_workspace_bytes(self.vid, T, B, C, S)
        self.nll = torch.empty(B, dtype=torch.float32, device=dev)
        self.grad = torch.empty(T, B, C, dtype=torch.float32, device=dev)
        self.one = torch.ones((), dtype=torch.float32, device=dev)
        self.scale = 1.0 / B_global
        # logits read once + gradient written once + targets + lengths + loss
        self.alg_bytes = 8 * T * B * C + self.target_bytes + 16 * B + 4
        self.cells = B * T * (2 * S + 1 if variant == "blank" else S)

    def new_workspace(self):
        return torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.dev)

    def fused(self, loss_ptr, ws, stream):
        x, lib = self.x, self.lib
        common = (self.il.data_ptr(), self.tl.data_ptr(), self.T, self.B, self.C, self.S)
        tail = (self.scale, self.scale, self.nll.data_ptr(), loss_ptr, self.grad.data_ptr(), ws.data_ptr(), stream)
        if self.variant == "noblank":
            rc = lib.ctc_amd_noblank_loss_grad(x.data_ptr(), x.stride(0), x.stride(1), self.tg.data_ptr(),
                                               int(self.tg.dtype == torch.int64), *common, *tail)
        elif self.variant == "binary":
            rc = lib.ctc_amd_binary_loss_grad(x.data_ptr(), x.stride(0), x.stride(1), self.tg.data_ptr(),
                                              *common, *tail)
        else:
            rc = lib.ctc_amd_blank_loss_grad(x.data_ptr(), x.stride(0), x.stride(1), self.tg.data_ptr(),
                                             int(self.tg.dtype == torch.int64), *common, 0, *tail)
        if rc:
            raise RuntimeError("fused launch failed: %d" % rc)

    def step(self, loss_ptr, ws, stream):
        self.fused(loss_ptr, ws, stream)
        rc = self.lib.ctc_amd_scale_grad(self.grad.data_ptr(), self.one.data_ptr(), self.grad.numel(), stream)
        if rc:
            raise RuntimeError("scale_grad launch failed: %d" % rc)


def cur_stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def oracle_step(wl, threads, want_grad=True, dtype=np.float32):
    from oracle import ctc_c
    from tests.helpers import np_
    x, tg, il, tl = (np_(t) for t in wl.host)
    fn = {"noblank": ctc_c.noblank_ctc, "binary": ctc_c.binary_ctc, "blank": ctc_c.blank_ctc}[wl.variant]
    return fn(x, tg, il, tl, dtype, threads=threads, want_grad=want_grad)


def cpu_baseline(wl, budget_s):
    """The oracle (C port of the reference arithmetic) on the host cores, bounded sample."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nthr = min(ncpu, 16)                                   # the 1-GPU box's CPU share
    out = {}
    for label, th in (("1", 1), ("all", nthr)):
        oracle_step(wl, th)                                # warm (thread pool, page faults)
        n, t0 = 0, time.perf_counter()
        while True:
            oracle_step(wl, th)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 200:
                break
        out[label] = (wl.B * n / el, th, n)
    best = max(out.values(), key=lambda v: v[0])
    return {"value": round(best[0], 2), "unit": "samples/s", "cores": best[1], "kind": "port",
            "sample": "%d full steps (loss+grad) of the same %s batch, oracle/ctc_oracle.c fp32, OpenMP over samples"
                      % (best[2], "B=%d T=%d C=%d S=%d" % (wl.B, wl.T, wl.C, wl.S)),
            "single_thread_value": round(out["1"][0], 2), "host_cores_visible": ncpu}


def torch_cpu_ctc_baseline(wl, budget_s):
    """torch.nn.functional.ctc_loss on the host cores beside the blank variant."""
    x, tg, il, tl = wl.host
    n, t0 = 0, time.perf_counter()
    while True:
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.ctc_loss(xr, tg, il, tl, blank=0, reduction="mean").backward()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": round(wl.B * n / el, 2), "unit": "samples/s", "threads": torch.get_num_threads(),
            "what": "torch.nn.functional.ctc_loss fwd+bwd on CPU, %d steps of the same batch" % n}


def _host_ext_loaded():
    from ctc_amd import functional as F
    return F._host_ext not in (False, None)


def eager_python_step(wl, iters=300):
    """The step through the Python surface, issued eagerly as the reference's loop does
    (train.py:427,444): CTCLoss.apply / blank_ctc_loss + loss.backward()."""
    import ctc_amd
    from ctc_amd import functional as F
    x = wl.x.clone().requires_grad_(True)

    def one():
        x.grad = None
        if wl.variant == "blank":
            loss, _ = F.blank_ctc_loss(x, wl.tg, wl.il, wl.tl)
        else:
            loss = ctc_amd.CTCLoss.apply(x, wl.tg, wl.il, wl.tl)
        loss.backward()
    def floor():                                             # the cheapest loss torch itself offers on the same tensor:
        x.grad = None                                        # what one eager forward + backward() costs before any CTC
        x.sum().backward()

    def forward_only():
        with torch.no_grad():
            if wl.variant == "blank":
                F.blank_ctc_loss(x, wl.tg, wl.il, wl.tl)
            else:
                ctc_amd.CTCLoss.apply(x, wl.tg, wl.il, wl.tl)

    def timed(fn):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6
    return timed(one), timed(floor), timed(forward_only)


def main():
    a = parse()
    if a.watchdog_seconds > 0:
        import threading

        def _give_up():
            print("bench: not finished after %d s, giving up" % a.watchdog_seconds, file=sys.stderr, flush=True)
            os._exit(3)
        wd = threading.Timer(a.watchdog_seconds, _give_up)
        wd.daemon = True
        wd.start()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    if local >= ndev and a.backend == "nccl":
        raise SystemExit("rank %d has no GPU (%d visible): RCCL needs one GPU per rank" % (local, ndev))
    local %= max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    import datetime
    coll = world > 1 or a.rehearse_collective               # the step is followed by a loss all-reduce
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # (a stuck collective ends the run after three minutes instead of hanging it)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))

    variant = a.variant
    K = a.steps if a.steps is not None else (500 if variant != "blank" else 30)
    W = a.warmup if a.warmup is not None else (50 if variant != "blank" else 5)
    if a.scaling == "strong":
        from ctc_amd.distributed import shard_bounds
        Bg = a.global_batch
        lo, hi = shard_bounds(Bg, rank, world)
        wl = Workload(variant, Bg, Bg, dev, seed=0, lo=lo, hi=hi,
                      name=workload_name(variant, hi - lo, Bg, world, "strong"))
    else:
        Bl = a.batch or WORKLOADS[variant][4]
        Bg = Bl * world
        wl = Workload(variant, Bl, Bg, dev, seed=rank)
    B = wl.B

    # N > 1: ONE all-reduce of the scalar loss per step is the primary figure (BASELINE north_star / configs[3]);
    # `--loss-bucket M` amortises it over M steps and is what the default run reports as `bucketed`.  (Measured on one GPU
    # with stand-in collective kernels, DESIGN.md section 5: a collective kernel node on a SECOND stream costs a captured
    # graph ~18 us per step in cross-stream dependencies and, resident beside a B = #CUs loss launch, takes whole CUs away
    # from it -- hence the same-stream placement below.)
    bucket = (a.loss_bucket if a.loss_bucket is not None else 1) if coll else None
    per_step_collective = coll and bucket == 1
    # one all-reduce per step INSIDE the hipGraph (asynchronous: it runs on RCCL's stream beside the next step's kernel)
    in_graph = per_step_collective and a.launch == "graph" and a.collective_launch == "graph" and a.backend == "nccl"
    launch = "graph" if in_graph else ("eager" if per_step_collective else a.launch)
    M = max(1, min(a.graph_steps, K)) if launch == "graph" else 1
    if bucket is None or (launch == "graph" and coll and not in_graph):
        bucket = M if launch == "graph" else (bucket or 1)

    # ---- the step sequence, eager or captured into hipGraphs (two, alternating, so that an
    # all-reduce of one loss ring can overlap the replay that fills the other)
    rings = [torch.zeros(max(M, bucket, a.graph_steps), dtype=torch.float32, device=dev) for _ in range(2)]
    ws = wl.new_workspace()
    wl.step(rings[0].data_ptr(), ws, cur_stream(dev))       # first touch outside any capture
    torch.cuda.synchronize()

    side = torch.cuda.Stream(dev)                            # the collectives are ordered behind this stream
    occ_sink = torch.zeros(4, dtype=torch.int32, device=dev)

    def drain():
        """no NCCL work outstanding and nothing in flight on the device: a capture must not begin while the process
        group's watchdog thread still polls the events of earlier asynchronous all-reduces (a hipEventQuery from
        another thread invalidates a capture in the global error mode -- round 3's `scale_grad launch failed: 901`)"""
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()
        if coll and state.get("eager_works"):               # works the watchdog may still hold: it retires finished ones
            time.sleep(0.6)                                 # every ~100 ms; after that it has no event left to query
            state["eager_works"] = False

    def capture(m, collective=False, gate=False, occupant=0, placement=None):
        """m steps per graph.  collective: one all-reduce of the step's loss per step -- placement "same": a blocking
        all-reduce enqueued on the capture stream itself behind scale_grad; "side": asynchronous, issued from a side
        stream that forks off behind the step's kernels, optionally behind the collective gate (so that RCCL's kernel is
        dispatched after the NEXT step's loss launch has filled the chip) and, when rehearsing on one GPU, behind
        `occupant` stand-in workgroups with the collective kernel's footprint."""
        placement = placement or a.collective_stream
        if occupant or gate:
            placement = "side"                              # (the stand-in kernel and the gate belong to the side stream)
        drain()
        gs = []
        for r in rings:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                main = torch.cuda.current_stream(dev)
                s = main.cuda_stream
                works = []
                for j in range(m):
                    wl.step(r.data_ptr() + 4 * j, ws, s)
                    if collective and placement == "same":
                        dist.all_reduce(r[j:j + 1], op=dist.ReduceOp.SUM)
                    elif collective:
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            if gate and j < m - 1:          # (only when a loss launch follows in this graph: the gate is bounded, not free)
                                rc = wl.lib.ctc_amd_collective_gate(ws.data_ptr(), wl.B, 30, side.cuda_stream)
                                assert rc == 0, rc
                            if occupant:
                                occ_lib.coresident_launch(occupant, a.occupant_us, occ_sink.data_ptr(), side.cuda_stream)
                            works.append(dist.all_reduce(r[j:j + 1], op=dist.ReduceOp.SUM, async_op=True))
                for w in works:                             # (joins RCCL's stream back into the capture)
                    w.wait()
                if collective and placement == "side":
                    main.wait_stream(side)
            gs.append(g)
        torch.cuda.synchronize()
        return gs

    def try_capture(what, *args, **kw):
        """a capture that fails (any exception) is reported in one line and returns None: the caller falls back"""
        try:
            return capture(*args, **kw)
        except Exception as e:                              # noqa: BLE001
            print("bench: capture of %s failed (%s: %s)" % (what, type(e).__name__, str(e).splitlines()[0][:200]),
                  file=sys.stderr, flush=True)
            try:
                torch.cuda.synchronize()
            except Exception:                               # noqa: BLE001
                pass
            return None

    pending = [None, None]
    state = {"i": 0, "eager_works": False}
    if coll:                                                # communicator set up outside any capture
        dist.all_reduce(rings[0][:1], op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        rings[0].zero_()
        state["eager_works"] = True
    graphs = []
    occ_lib = None
    n_occ = a.occupant or 0                                 # (only on request: the default rehearsal runs exactly what N > 1 runs)
    if n_occ:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import coresident
        occ_lib = coresident.load_occupant()
    use_gate = bool(a.collective_gate) and variant != "blank"
    capture_note = None
    if in_graph:
        graphs = try_capture("the step with its all-reduce", M, collective=True, gate=use_gate, occupant=n_occ)
        if graphs is not None:
            try:
                graphs[0].replay()
                torch.cuda.synchronize()
            except Exception as e:                          # noqa: BLE001
                print("bench: replay of the captured collective failed (%s)" % str(e).splitlines()[0][:200], file=sys.stderr)
                graphs = None
        if graphs is None:                                  # the eager path still works: steps and all-reduces one by one
            capture_note = "collective capture failed: steps and all-reduces issued eagerly (host-bound)"
            in_graph, launch, M, graphs = False, "eager", 1, []
            bucket = 1
    elif launch == "graph":
        graphs = try_capture("the step", M)
        if graphs is None:
            capture_note = "graph capture failed: steps issued eagerly"
            launch, M, graphs = "eager", 1, []
            bucket = bucket or 1
    # Every other graph of this run is captured NOW, before the measurement phase issues its first eager all-reduce:
    # ProcessGroupNCCL's watchdog thread polls the events of eager works, and an event query from another thread while a
    # capture is in progress aborts the process on this stack even in the thread-local capture mode (seen once in round 3
    # and once in round 4, both times in a capture that FOLLOWED eager all-reduces).  Works issued inside a capture are
    # not handed to the watchdog, so captured collectives are safe; drain() additionally waits out the watchdog's period.
    gp = gb = None
    Mb = max(1, min(a.graph_steps, K))
    want_per_step = coll and not per_step_collective and a.launch == "graph" and a.backend == "nccl" and launch == "graph"
    want_bucketed = per_step_collective and a.launch == "graph"
    if want_per_step:
        gp = try_capture("the per-step all-reduce (secondary figure)", M, collective=True, gate=use_gate, occupant=0)
    if want_bucketed:
        gb = try_capture("the bucketed form (secondary figure)", Mb)
    co_graphs = {}
    co_specs = (("gated_us_per_step", dict(collective=True, gate=True, occupant=n_occ)),
                ("ungated_us_per_step", dict(collective=True, gate=False, occupant=n_occ)),
                ("allreduce_only_us_per_step", dict(collective=True, gate=False, occupant=0, placement="side")),
                ("allreduce_same_stream_us_per_step", dict(collective=True, gate=False, occupant=0, placement="same")),
                ("no_collective_us_per_step", dict(collective=False)))
    if in_graph and n_occ and world == 1:
        for label, kw in co_specs:
            co_graphs[label] = try_capture(label, M, **kw)

    def run_steps(n, launch=launch, graphs=graphs, M=M, bucket=bucket, graph_collective=in_graph):
        """exactly n steps; N>1: loss contributions all-reduced every `bucket` steps (graph_collective: the graphs
        carry one asynchronous all-reduce per step themselves)"""
        done = 0
        while done < n:
            k = state["i"] & 1
            if pending[k] is not None:                      # ring k is about to be overwritten
                pending[k].wait()
                pending[k] = None
            replayed = launch == "graph" and n - done >= M
            if replayed:
                graphs[k].replay()
                m = M
            else:
                m = min(bucket, n - done) if (launch == "eager" or graph_collective) else n - done
                s = cur_stream(dev)
                for j in range(m):
                    wl.step(rings[k].data_ptr() + 4 * j, ws, s)
            if coll and not (replayed and graph_collective):
                pending[k] = dist.all_reduce(rings[k][:max(m, 1)], op=dist.ReduceOp.SUM, async_op=True)
                state["eager_works"] = True
            state["i"] += 1
            done += m
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, **kw):
        """exactly n steps between two fences; max over ranks -> seconds"""
        fence()
        t0 = time.perf_counter()
        run_steps(n, **kw)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    run_steps(W)
    # K steps are timed in one bracketed region.  A short K (the driver uses 20) is one or two graph
    # replays, i.e. mostly launch latency of the replay itself: the region is then repeated and the
    # median taken (>= 50 timed iterations), every repeat timing exactly K steps.
    repeats = 1 if K >= 100 else min(15, max(3, -(-150 // max(K, 1))))
    els = sorted(timed(K) for _ in range(repeats))
    el = els[len(els) // 2]

    # secondary figures for N>1: the same steps with the loss all-reduce amortised over a graph replay (`bucketed`, when
    # the primary run has one all-reduce per step), or with one all-reduce per step captured into the graph
    # (`per_step_collective`, when the primary run is the bucketed default)
    per_step = None
    if want_per_step:
        if gp is None:
            per_step = {"error": "capture failed"}
        else:
            run_steps(M, launch="graph", graphs=gp, M=M, bucket=1, graph_collective=True)
            npst = max(M, min(K, 200) // M * M)
            elp = timed(npst, launch="graph", graphs=gp, M=M, bucket=1, graph_collective=True)
            per_step = {"value": round(B_total(world, wl, a) * npst / elp, 1), "unit": "samples/s", "steps": npst,
                        "ms_per_step": round(elp / npst * 1e3, 6), "launch": "graph", "graph_steps": M,
                        "loss_allreduce_bucket": 1, "collective_launch": "in the hipGraph, one all-reduce per step (%s stream)" % a.collective_stream}
    bucketed = None
    if want_bucketed:
        if gb is None:
            bucketed = {"error": "capture failed"}
        else:
            run_steps(Mb, launch="graph", graphs=gb, M=Mb, bucket=Mb, graph_collective=False)
            nb = max(Mb, K // Mb * Mb)
            elb = timed(nb, launch="graph", graphs=gb, M=Mb, bucket=Mb, graph_collective=False)
            bucketed = {"value": round(B_total(world, wl, a) * nb / elb, 1), "unit": "samples/s", "steps": nb,
                        "ms_per_step": round(elb / nb * 1e3, 6), "launch": "graph", "graph_steps": Mb,
                        "loss_allreduce_bucket": Mb, "what": "the same steps, the losses of one graph replay in ONE all-reduce"}

    # rehearsal on one GPU: the same graph with the stand-in collective kernel, gated and not, and without a collective
    coresident = None
    if co_graphs:
        coresident = {"occupant": "%d workgroups with rcclGenericKernel's footprint (256 threads, 19744 B LDS, 280 registers "
                                  "per lane), resident %.0f us, where each step's all-reduce kernel would run" % (n_occ, a.occupant_us)}
        for label, _kw in co_specs:
            gx = co_graphs.get(label)
            if gx is None:
                coresident[label] = None
                continue
            nb = max(M, K // M * M)
            run_steps(M, launch="graph", graphs=gx, M=M, bucket=M, graph_collective=True)
            elx = sorted(timed(nb, launch="graph", graphs=gx, M=M, bucket=M, graph_collective=True) for _ in range(3))[1]
            coresident[label] = round(elx / nb * 1e6, 3)

    # ---- dominant kernel: per-launch duration from HIP events on the launch stream
    n_ev = 200 if variant != "blank" else 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    s = cur_stream(dev)
    for e0, e1 in ev:
        e0.record()
        wl.fused(rings[0].data_ptr(), ws, s)
        e1.record()
    torch.cuda.synchronize()
    per = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)          # us
    kern_us = float(np.mean(per))
    e0, e1 = ev[0]
    e0.record()
    for _ in range(n_ev):
        wl.fused(rings[0].data_ptr(), ws, s)
    e1.record()
    torch.cuda.synchronize()
    b2b_us = e0.elapsed_time(e1) * 1e3 / n_ev

    out = None
    if rank == 0:
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("%s_B%d" % (variant, B), {}).get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = "profiles/traffic.json (rocprofv3 --pmc passes of tools/profile.sh, not this run)"
        # average launch duration: n_ev launches between two HIP events on the launch stream (this
        # includes the inter-launch gap and is what rocprofv3 --stats reports, profiles/*.md); the
        # per-launch bracketed figure carries ~2 us of event overhead and is kept for reference
        achieved = wl.alg_bytes / (b2b_us * 1e-6) / 1e9
        sps = Bg * K / el
        # BASELINE.md section 4's own definition: algorithmic bytes of this rank's batch / the whole timed step
        # (fused launch + the scale_grad boundary of backward [+ the collective]) / peak
        achieved_step = wl.alg_bytes / (el / K) / 1e9
        out = {
            "metric": "ctc_samples_per_sec", "value": round(sps, 1), "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 6),
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name, "variant": variant, "per_gpu_batch": B, "global_batch": Bg,
                       "T": wl.T, "C": wl.C, "S": wl.S, "parallelism": "dp%d (batch-sharded)" % world,
                       "collective": (a.backend if coll else None),
                       "collective_launch": (("in the hipGraph, one all-reduce per step on the %s stream" % ("launch" if (a.collective_stream == "same" and not n_occ and not use_gate) else "side") if in_graph else
                                              "eager, one asynchronous all-reduce per step" if per_step_collective else
                                              "one all-reduce per %d steps" % bucket) if coll else None),
                       "launch": launch, "graph_steps": M if launch == "graph" else None,
                       "loss_allreduce_bucket": bucket if coll else None,
                       "timed_repeats_median_of": repeats,
                       "step": "fused loss+grad launch + scale_grad launch (loss.backward(), grad_out=1)"},
            "lattice_cells_per_sec": round(wl.cells * (Bg / B) * K / el, 1),
            "lattice_cells_per_sec_2Sp1": round(Bg * wl.T * (2 * wl.S + 1) * K / el, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "achieved_step": round(achieved_step, 1), "frac_step": round(achieved_step / HBM_PEAK_GBS, 4),
                         "frac_step_what": "algorithmic bytes / ms_per_step / peak: the fused launch AND the scale_grad "
                                           "launch of backward (a kernel boundary that moves no data when grad_out == 1)",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "%s fused loss+grad" % variant, "kernel_us_avg": round(b2b_us, 3),
                         "kernel_us_event_bracketed_avg": round(kern_us, 3),
                         "kernel_us_event_bracketed_median": round(per[len(per) // 2], 3),
                         "algorithmic_bytes_per_launch": wl.alg_bytes},
        }
        if bucketed is not None:
            out["bucketed"] = bucketed
        if per_step is not None:
            out["per_step_collective"] = per_step
        if coresident is not None:
            out["coresident"] = coresident
        if coll:
            out["config"]["collective_gate"] = bool(use_gate and in_graph)
            out["config"]["workload"] += "; loss all-reduce every %s" % ("step" if bucket == 1 else "%d steps" % bucket)
        if capture_note:
            out["config"]["note"] = capture_note
        # (before anything multi-threaded runs on the host: the checker's OpenMP workers keep spinning for a while after
        # their last parallel region and the autograd engine's thread wake-ups then take 2-3x as long)
        eager = eager_python_step(wl, 300 if variant != "blank" else 20) if (world == 1 and not a.no_eager_python) else None
        # parity of THIS batch against the float64 oracle (untimed)
        ref = oracle_step(wl, threads=min(16, os.cpu_count() or 1), dtype=np.float64)
        torch.cuda.synchronize()
        nll_dev = wl.nll.cpu().numpy().astype(np.float64)
        out["parity"] = {"checker": "oracle/ctc_oracle.c float64",
                         "max_rel_err_nll": float((np.abs(nll_dev - ref["nll"]) / np.maximum(1.0, np.abs(ref["nll"]))).max()),
                         "max_abs_err_nll": float(np.abs(nll_dev - ref["nll"]).max()),
                         "max_abs_err_grad": float(np.abs(wl.grad.cpu().numpy() - ref["grad"] * (B / Bg)).max())}
        del ref
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, a.cpu_seconds)
            if variant == "blank":
                out["cpu_baseline_torch"] = torch_cpu_ctc_baseline(wl, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        if eager is not None:
            step_us, floor_us, fwd_us = eager
            out["eager_python"] = {"us_per_step": round(step_us, 2),
                                   "launch": "eager-python",
                                   "what": "CTCLoss.apply(...) + loss.backward() issued eagerly (train.py:427,444), host-inclusive",
                                   "torch_floor_us": round(floor_us, 2),
                                   "torch_floor_what": "x.sum().backward() on the same tensor, same loop: the autograd engine's own cost per eager step",
                                   "forward_only_us": round(fwd_us, 2),
                                   "autograd_node": "C++ (ctc_amd/csrc/autograd_ext.cpp)" if (variant != "blank" and _host_ext_loaded())
                                   else "python (ctc_amd/functional.py)"}
    if coll:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def B_total(world, wl, a):
    return a.global_batch if a.scaling == "strong" else wl.B * world


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:                              # noqa: BLE001 -- a failed run says why in one line and exits non-zero
        import traceback
        traceback.print_exc()
        print("bench: FAILED (%s: %s)" % (type(e).__name__, str(e).splitlines()[0][:300] if str(e) else ""), file=sys.stderr, flush=True)
        os._exit(1)
