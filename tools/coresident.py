#!/usr/bin/env python3
"""Diagnostic: what a collective kernel that is RESIDENT on k CUs does to the one-workgroup-per-CU loss launch.

    python tools/coresident.py [--variant noblank] [--batch 256]

A second stream holds k workgroups with the footprint of RCCL's collective kernel (tools/micro/coresident.hip:
256 threads, 19.7 KB of LDS, 280 registers per lane -- it cannot share a CU with the loss kernel's workgroup)
resident for ~60 us; the fused loss+gradient launch is timed with HIP events on its own stream
  order "collective first":  occupant enqueued, then the loss launch      (the collective got its CUs first)
  order "loss first":        the loss launch enqueued, then the occupant  (the occupant waits for a free CU)
  order "gated":             occupant behind ctc_amd's arrival gate, see ctc_amd_gate_* in the header
Prints one JSON line (median us per launch for k = 0, 1, 2, 8).  Not on the product path.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def load_occupant():
    from ctc_amd import build
    so = build.build_occupant()           # prebuilt by __graft_entry__.build(); compiled here only if missing / stale
    lib = ctypes.CDLL(so)
    lib.coresident_launch.restype = ctypes.c_int
    lib.coresident_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def measure(wl, occ, k, order, iters=60, hold_us=60.0, gate=None):
    dev = wl.dev
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ws = wl.new_workspace()
    loss = torch.zeros(4, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    per = []
    for it in range(iters + 10):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if k > 0 and order == "collective first":
            occ.coresident_launch(k, hold_us, sink.data_ptr(), sb.cuda_stream)
        if k > 0 and order == "gated":                       # enqueued first, but behind the arrival gate of THIS launch
            rc = wl.lib.ctc_amd_collective_gate(ws.data_ptr(), wl.B, 100, sb.cuda_stream)
            assert rc == 0, rc
            occ.coresident_launch(k, hold_us, sink.data_ptr(), sb.cuda_stream)
        with torch.cuda.stream(sa):
            e0.record()
            wl.fused(loss.data_ptr(), ws, sa.cuda_stream)
            e1.record()
        if k > 0 and order == "loss first":
            occ.coresident_launch(k, hold_us, sink.data_ptr(), sb.cuda_stream)
        torch.cuda.synchronize()
        if it >= 10:
            per.append(e0.elapsed_time(e1) * 1e3)
    per.sort()
    return per[len(per) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="noblank")
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch or bench.WORKLOADS[a.variant][4]
    wl = bench.Workload(a.variant, B, B, dev, 0)
    occ = load_occupant()
    out = {"workload": wl.name, "occupant": "256 threads, 19744 B LDS, 280 registers per lane (rcclGenericKernel's footprint), "
                                           "resident ~60 us", "us_per_launch_event_bracketed_median": {}}
    for order in ("collective first", "loss first", "gated"):
        row = {}
        for k in (0, 1, 2, 8):
            row["k=%d" % k] = round(measure(wl, occ, k, order), 2)
        out["us_per_launch_event_bracketed_median"][order] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
