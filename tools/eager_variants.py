#!/usr/bin/env python3
"""Host-inclusive cost of one eager step through the Python surface in the forms a training loop may use it
(config 2: B=256 T=150 C=158 S<=20): the reference's own call, lengths on the host, a strided logits view, a scaled loss,
no gradient."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctc_amd  # noqa: E402
from tests.helpers import synth_noblank  # noqa: E402

dev = torch.device("cuda:0")
x, lab, Tb, L = synth_noblank(0, 150, 256, 158, 20)
xs = x.to(dev).requires_grad_(True)
lab, Tbd, Ld = lab.to(dev), Tb.to(dev), L.to(dev)
big = torch.randn(150, 512, 158, device=dev)
view = big[:, ::2].detach().requires_grad_(True)
crit = ctc_amd.NoBlankCTC()


def timed(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def step(xin, il, tl, scale=None, module=False):
    def f():
        xin.grad = None
        loss = crit(xin, lab, il, tl) if module else ctc_amd.CTCLoss.apply(xin, lab, il, tl)
        if scale is not None:
            loss = loss * scale
        loss.backward()
    return f


def fwd_only():
    with torch.no_grad():
        ctc_amd.CTCLoss.apply(xs, lab, Tbd, Ld)


def floor():
    xs.grad = None
    xs.sum().backward()


rows = [("torch floor: x.sum().backward()", floor),
        ("CTCLoss.apply + backward (train.py:427,444)", step(xs, Tbd, Ld)),
        ("NoBlankCTC module + backward", step(xs, Tbd, Ld, module=True)),
        ("... lengths on the host", step(xs, Tb, L)),
        ("... strided logits view (every second sample of a larger batch)", step(view, Tbd, Ld)),
        ("... (loss * 0.25).backward()  (gradient accumulation)", step(xs, Tbd, Ld, scale=0.25)),
        ("forward only, no_grad", fwd_only)]
for name, fn in rows:
    print("%-70s %7.1f us per step" % (name, timed(fn)))
