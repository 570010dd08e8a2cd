#!/usr/bin/env python3
"""Diagnostic: blank-CTC loss+gradient call time over sequence lengths, for choosing between the persistent
launch and the three launches: each length is timed with ctc_amd_blank_set_schedule(1) and (0).  Not part of the product path.
usage: tools/blank_sweep.py [--shape B,C,S] T..."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import ctc_amd  # noqa: E402

argv = sys.argv[1:]
B, C, S = 64, 1000, 100
if argv and argv[0] == "--shape":
    B, C, S = (int(v) for v in argv[1].split(","))
    argv = argv[2:]
Ts = [int(v) for v in argv] or [128, 256, 512, 1000, 2000]


def timed(T):
    bench.WORKLOADS["blank"] = ("sweep", T, C, S, B, "sweep")
    wl = bench.Workload("blank", B, B, torch.device("cuda:0"), 0)
    ws = wl.new_workspace()
    loss = torch.zeros(4, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        wl.fused(loss.data_ptr(), ws, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        wl.fused(loss.data_ptr(), ws, s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for T in Ts:
    t = {}
    for mode in ("1", "0"):
        ctc_amd.set_blank_schedule(int(mode))
        t[mode] = timed(T)
    print("B=%d C=%d S=%d T=%5d: persistent launch %8.1f us, three launches %8.1f us per call" % (B, C, S, T, t["1"], t["0"]))
ctc_amd.set_blank_schedule(-1)
