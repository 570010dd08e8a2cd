#!/usr/bin/env python3
"""Wall-clock cost of one step through the PYTHON surface (CTCLoss.apply + loss.backward()),
i.e. including the autograd engine, allocation of the gradient buffer and ctypes overhead --
next to the C-ABI step that bench.py reports.  Eager and hipGraph-captured."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("CTC_AMD_PKG_ROOT"):                     # A/B against another copy of the package
    sys.path.insert(0, os.environ["CTC_AMD_PKG_ROOT"])
import ctc_amd  # noqa: E402
from tests.helpers import synth_noblank  # noqa: E402

dev = torch.device("cuda:0")
x, lab, Tb, L = synth_noblank(0, 150, 256, 158, 20)
xs = x.to(dev).requires_grad_(True)
lab, Tb, L = lab.to(dev), Tb.to(dev), L.to(dev)


def step():
    xs.grad = None
    loss = ctc_amd.CTCLoss.apply(xs, lab, Tb, L)
    loss.backward()
    return loss


for _ in range(50):
    step()
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / n * 1e6

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
xs.grad = None
with torch.cuda.graph(g):
    for _ in range(20):
        step()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(25):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / (25 * 20) * 1e6
print("autograd path, B=256 T=150 C=158 S<=20: eager %.1f us/step (host-bound), hipGraph replay %.1f us/step" % (eager, graph))
