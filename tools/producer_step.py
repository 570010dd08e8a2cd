#!/usr/bin/env python3
"""What the recurrence of LSTM_cell.forward (LSTM.py:44-51) costs on the device at the reference's sizes: torch's own
nn.LSTMCell loop (the reference), one fused launch per frame (ctc_amd_lstm_cell_step), one launch for all frames
(ctc_amd_lstm_series).  The per-frame feature head (Linear + BatchNorm + ReLU + Dropout) is torch's in all three and
is not timed."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for (T, B, H) in [(150, 10, 33), (150, 64, 33), (150, 256, 33), (150, 10, 38)]:
    cell = torch.nn.LSTMCell(H, H).to(dev)
    v_all = torch.randn(T, B, H, device=dev)
    h0, c0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    p = (cell.weight_ih.detach(), cell.weight_hh.detach(), cell.bias_ih.detach(), cell.bias_hh.detach())

    def torch_loop():
        with torch.no_grad():
            h, c = h0, c0
            out = torch.empty(T, B, H, device=dev)
            for t in range(T):
                h, c = cell(v_all[t], (h, c))
                out[t] = h
        return out

    def step_loop():
        out = torch.empty(T, B, H, device=dev)
        h, c = h0, c0
        for t in range(T):
            h, c, _ = ctc_amd.lstm_cell_step(v_all[t], h, c, *p, out[t])
        return out

    def one_launch():
        return ctc_amd.lstm_series(v_all, h0, c0, *p)[0]

    from ctc_amd import producer
    va = v_all.clone().requires_grad_(True)

    def torch_train():
        cell.zero_grad(); va.grad = None
        h, c, rows = h0, c0, []
        for t in range(T):
            h, c = cell(va[t], (h, c))
            rows.append(h)
        torch.stack(rows).sum().backward()

    def hip_train():
        cell.zero_grad(); va.grad = None
        producer._SeriesFn.apply(va, h0, c0, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, H, producer.PAD_LOGIT).sum().backward()

    print("    forward + backward: torch nn.LSTMCell loop + autograd %9.1f us | one launch each way + three GEMMs %7.1f us"
          % (timed(torch_train, 10), timed(hip_train, 10)))
    err = (one_launch() - torch_loop()).abs().max().item()
    print("T=%d B=%3d H=%d: torch nn.LSTMCell loop %8.1f us | one fused launch per frame %8.1f us | one launch %7.1f us   (max |diff| to torch %.1e)"
          % (T, B, H, timed(torch_loop), timed(step_loop), timed(one_launch), err))

# ---- the head (Linear + BatchNorm1d + ReLU + Dropout, LSTM.py:8-18) for all frames: torch's layers frame by frame (the
# reference) against ctc_amd_head_forward (one launch), eval mode and train mode (p = 0.3), forward and forward + backward
for (T, B, K, C) in [(150, 10, 1024, 33), (150, 64, 1024, 33), (150, 256, 1024, 158)]:
    lin, bn = torch.nn.Linear(K, C).to(dev), torch.nn.BatchNorm1d(C).to(dev)
    feat = torch.randn(T, B, K, device=dev)
    mask = torch.nn.functional.dropout(torch.ones(T, B, C, device=dev), 0.3, True)
    for train in (False, True):
        bn.train(train)

        def torch_head():
            with torch.no_grad():
                return torch.stack([torch.relu(bn(lin(feat[t]))) for t in range(T)]) * mask

        def hip_head():
            with torch.no_grad():
                rm, rv = (None, None) if train else (bn.running_mean, bn.running_var)
                return ctc_amd.head_forward(feat, lin.weight, lin.bias, bn.weight, bn.bias, rm, rv, bn.eps, mask)[0]

        fr = feat.clone().requires_grad_(True)

        def torch_head_train():
            lin.zero_grad(); bn.zero_grad(); fr.grad = None
            (torch.stack([torch.relu(bn(lin(fr[t]))) for t in range(T)]) * mask).sum().backward()

        def hip_head_train():
            lin.zero_grad(); bn.zero_grad(); fr.grad = None
            rm, rv = (None, None) if train else (bn.running_mean, bn.running_var)
            producer._HeadFn.apply(fr, lin.weight, lin.bias, bn.weight, bn.bias, rm, rv, bn.eps, mask)[0].sum().backward()

        err = (hip_head() - torch_head()).abs().max().item()
        print("head T=%d B=%3d K=%d C=%3d %s: torch layers per frame %8.1f us | one launch %7.1f us | fwd+bwd %9.1f / %8.1f us   (max |diff| %.1e)"
              % (T, B, K, C, "train" if train else "eval ", timed(torch_head, 10), timed(hip_head, 10), timed(torch_head_train, 5),
                 timed(hip_head_train, 5), err))
