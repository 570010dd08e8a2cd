#!/usr/bin/env python3
"""Diagnostic: where the host time of an eager CTCLoss.apply + backward() step goes (cProfile)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctc_amd  # noqa: E402
from tests import helpers  # noqa: E402

dev = torch.device("cuda:0")
x, tg, il, tl = (t.to(dev) for t in helpers.synth_noblank(0, 150, 256, 158, 20))
x.requires_grad_(True)


def one():
    x.grad = None
    loss = ctc_amd.CTCLoss.apply(x, tg, il, tl)
    loss.backward()


def fwd_only():
    with torch.no_grad():
        ctc_amd.CTCLoss.apply(x, tg, il, tl)


for _ in range(50):
    one()
torch.cuda.synchronize()
for name, fn in (("fwd+bwd", one), ("fwd only (no_grad)", fn2 := fwd_only)):
    t0 = time.perf_counter()
    for _ in range(500):
        fn()
    torch.cuda.synchronize()
    print("%s: %.1f us per step" % (name, (time.perf_counter() - t0) / 500 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    one()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
