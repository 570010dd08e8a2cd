#!/usr/bin/env python3
"""Random shapes through the persistent no-blank launch (B > 2 #CUs) against the same samples run 200 at a time (the
one-sample-per-workgroup launch): per-sample nll bit for bit, gradient rows to 1e-9 (they differ by the two launches' 1/B).
    python tools/ps_sweep.py [seed [shapes [big]]]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctc_amd
from tests.helpers import synth_noblank

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
big = len(sys.argv) > 3 and sys.argv[3] == "big"
bad = 0
for it in range(n):
    T = int(rng.integers(4, 169)); C = 2 * int(rng.integers(1, 97)); S = int(rng.integers(1, min(31, T) + 1)); B = int(rng.integers(513, 1400))
    if big:                                                  # logits + gradient beyond the memory-side cache: non-temporal stores, NX = 2
        T = int(rng.integers(140, 169)); C = 2 * int(rng.integers(75, 97)); S = int(rng.integers(1, 32)); B = int(rng.integers(1300, 1700))
    tp = T + 19
    while tp % 4 != 2:
        tp += 1
    rp = 32 * ((C + 31) // 32)
    if 3 * (S + 1) * tp * 8 + 14 * 4 * rp * 4 + 1024 > 160 * 1024 or 4 * ((C + 31) // 32) > 12 * 2:
        print("T=%3d B=%4d C=%3d S=%2d: skipped (lattice + tiles beyond 160 KB of LDS: not the four-rows-per-wave kernel)" % (T, B, C, S))
        continue
    x, lab, Tb, L = synth_noblank(1000 + it, T, B, C, S, var_T=True)
    for b in range(3, B, 131):                               # a few samples without alignment
        L[b] = min(S, T); Tb[b] = max(1, int(L[b]) - 1)
    def run(xs, labs, tbs, ls):
        xd = xs.to(dev).requires_grad_(True)
        loss, nll = ctc_amd.noblank_ctc_loss(xd, labs.to(dev), tbs.to(dev), ls.to(dev))
        loss.backward(); torch.cuda.synchronize()
        return nll.cpu().numpy(), xd.grad.cpu().numpy()
    nll, g = run(x, lab, Tb, L)
    ok = True
    for lo in range(0, B, 200):
        hi = min(B, lo + 200)
        nc, gc = run(x[:, lo:hi], lab[lo:hi], Tb[lo:hi], L[lo:hi])
        same = np.array_equal(nc, nll[lo:hi], equal_nan=True)
        err = np.abs(gc * ((hi - lo) / B) - g[:, lo:hi]).max()
        if not same or not err <= 1e-9:
            ok = False
            print("  MISMATCH T=%d B=%d C=%d S=%d chunk %d: nll same %s, grad err %.3e" % (T, B, C, S, lo, same, err))
    bad += 0 if ok else 1
    print("T=%3d B=%4d C=%3d S=%2d: %s" % (T, B, C, S, "ok" if ok else "FAILED"), flush=True)
print("%d shapes, %d failed" % (n, bad))
sys.exit(1 if bad else 0)
