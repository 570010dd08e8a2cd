#!/usr/bin/env python3
"""Throughput of the CTC loss hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--variant noblank|binary|blank]

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident input:
loss AND the full input gradient (what `loss = ctc_loss(...); loss.backward()` costs in
the reference, train.py:427,444), issued through the C ABI exactly as the autograd
Function issues it: one fused `*_loss_grad` launch plus the `scale_grad` launch of
backward (upstream gradient 1.0).  Default workload = BASELINE configs[1]:
NoBlankCTC, B=256 per GPU, T=150, C=158, S<=20, fp32.

N>1 (launched by torch.distributed.run, one rank per GPU): every rank runs the same
per-GPU batch (weak scaling, global batch 256*N, gradient scale 1/B_global) and the
per-step loss contributions are summed across ranks with RCCL all-reduce, `--loss-bucket`
steps per collective (1 = one all-reduce per step).

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
WORKLOADS = {                  # variant -> (name, T, C, S, default per-GPU batch)
    "noblank": ("NoBlankCTC B=256 T=150 C=158 S<=20 fp32 (BASELINE configs[1])", 150, 158, 20, 256),
    "binary": ("NoBlankBinaryCTC B=256 T=150 C=158 S<=20 fp32 (BASELINE configs[2])", 150, 158, 20, 256),
    "blank": ("blank-CTC B=64 T=2000 C=1000 S=100 fp32 (BASELINE configs[4])", 2000, 1000, 100, 64),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--variant", default="noblank", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the BASELINE config)")
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"],
                    help="graph: steps are replayed from a captured hipGraph; eager: one ctypes call per launch")
    ap.add_argument("--graph-steps", type=int, default=50, help="steps captured per hipGraph")
    ap.add_argument("--loss-bucket", type=int, default=None,
                    help="N>1: steps per RCCL all-reduce of the loss scalars (default = graph-steps, eager: 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 collective backend: nccl = RCCL over xGMI; gloo only to rehearse the multi-rank "
                         "path on a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU budget of the oracle baseline leg")
    return ap.parse_args()


class Workload:
    """Device-resident synthetic batch + the C-ABI call sequence of one step."""

    def __init__(self, variant, B, B_global, dev, seed):
        from ctc_amd import _lib
        from tests import helpers
        self.variant, self.B, self.dev = variant, B, dev
        self.name, self.T, self.C, self.S, _ = WORKLOADS[variant]
        T, C, S = self.T, self.C, self.S
        if variant == "noblank":
            x, tg, il, tl = helpers.synth_noblank(seed, T, B, C, S)
            self.target_bytes = 4 * B * S
        elif variant == "binary":
            x, tg, il, tl = helpers.synth_binary(seed, T, B, C, S)
            self.target_bytes = 4 * B * S * C
        else:
            x, tg, il, tl = helpers.synth_blank(seed, T, B, C, S)
            self.target_bytes = 8 * B * S
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


def main():
    a = parse()
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    variant = a.variant
    B = a.batch or WORKLOADS[variant][4]
    K = a.steps if a.steps is not None else (500 if variant != "blank" else 30)
    W = a.warmup if a.warmup is not None else (50 if variant != "blank" else 5)
    wl = Workload(variant, B, B * world, dev, seed=rank)
    M = max(1, min(a.graph_steps, K)) if a.launch == "graph" else 1
    bucket = a.loss_bucket or M

    # ---- the step sequence, eager or captured into hipGraphs (two, alternating, so that an
    # all-reduce of one loss ring can overlap the replay that fills the other)
    rings = [torch.zeros(max(M, bucket), dtype=torch.float32, device=dev) for _ in range(2)]
    ws = wl.new_workspace()
    wl.step(rings[0].data_ptr(), ws, cur_stream(dev))       # first touch outside any capture
    torch.cuda.synchronize()
    graphs = []
    if a.launch == "graph":
        for r in rings:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s = cur_stream(dev)
                for j in range(M):
                    wl.step(r.data_ptr() + 4 * j, ws, s)
            graphs.append(g)

    pending = [None, None]
    state = {"i": 0}

    def run_steps(n):
        """exactly n steps; N>1: loss contributions all-reduced every `bucket` steps"""
        done = 0
        while done < n:
            k = state["i"] & 1
            if pending[k] is not None:                      # ring k is about to be overwritten
                pending[k].wait()
                pending[k] = None
            if a.launch == "graph" and n - done >= M:
                graphs[k].replay()
                m = M
            else:
                m = min(bucket, n - done) if a.launch == "eager" else n - done
                s = cur_stream(dev)
                for j in range(m):
                    wl.step(rings[k].data_ptr() + 4 * j, ws, s)
            if world > 1:
                pending[k] = dist.all_reduce(rings[k][:max(m, 1)], op=dist.ReduceOp.SUM, async_op=True)
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

    run_steps(W)
    fence()
    t0 = time.perf_counter()
    run_steps(K)
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())

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
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("%s_B%d" % (variant, B), {}).get("hbm_bytes_per_launch")
        # average launch duration: n_ev launches between two HIP events on the launch stream (this
        # includes the inter-launch gap and is what rocprofv3 --stats reports, profiles/*.md); the
        # per-launch bracketed figure carries ~2 us of event overhead and is kept for reference
        achieved = wl.alg_bytes / (b2b_us * 1e-6) / 1e9
        sps = B * world * K / el
        out = {
            "metric": "ctc_samples_per_sec", "value": round(sps, 1), "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name, "variant": variant, "per_gpu_batch": B, "global_batch": B * world,
                       "T": wl.T, "C": wl.C, "S": wl.S, "parallelism": "dp%d (batch-sharded)" % world,
                       "collective": (a.backend if world > 1 else None),
                       "launch": a.launch, "graph_steps": M if a.launch == "graph" else None,
                       "loss_allreduce_bucket": bucket if world > 1 else None,
                       "step": "fused loss+grad launch + scale_grad launch (loss.backward(), grad_out=1)"},
            "lattice_cells_per_sec": round(wl.cells * world * K / el, 1),
            "lattice_cells_per_sec_2Sp1": round(B * wl.T * (2 * wl.S + 1) * world * K / el, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "%s fused loss+grad" % variant, "kernel_us_avg": round(b2b_us, 3),
                         "kernel_us_event_bracketed_avg": round(kern_us, 3),
                         "kernel_us_event_bracketed_median": round(per[len(per) // 2], 3),
                         "algorithmic_bytes_per_launch": wl.alg_bytes},
        }
        # parity of THIS batch against the float64 oracle (untimed)
        ref = oracle_step(wl, threads=min(16, os.cpu_count() or 1), dtype=np.float64)
        torch.cuda.synchronize()
        nll_dev = wl.nll.cpu().numpy().astype(np.float64)
        out["parity"] = {"checker": "oracle/ctc_oracle.c float64",
                         "max_rel_err_nll": float((np.abs(nll_dev - ref["nll"]) / np.maximum(1.0, np.abs(ref["nll"]))).max()),
                         "max_abs_err_nll": float(np.abs(nll_dev - ref["nll"]).max()),
                         "max_abs_err_grad": float(np.abs(wl.grad.cpu().numpy() - ref["grad"] * (1.0 / world)).max())}
        del ref
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
