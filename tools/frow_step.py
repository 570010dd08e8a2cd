#!/usr/bin/env python3
"""Every kernel of the SURVEY 8(f) rows at the reference's sizes, 50 calls each, for a rocprofv3 --kernel-trace --stats run
(tools/profile_frows.sh -> profiles/<round>_frow_kernel_stats.csv): the head and the LSTMCell recurrence of the producer
(forward and backward), best path and posteriors on both lattices, the target dedup, the label-smoothed loss."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctc_amd  # noqa: E402
from tests.helpers import synth_binary, synth_noblank  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("FROW_CALLS", "50"))
# the producer at the reference's sizes (opts.py: 1024 features, 33 classes, batch 10; 150 frames)
args = types.SimpleNamespace(extract_feat_dim=1024, v_class=33, batch_size=10, temporal=150)
model = ctc_amd.LSTM_cell(args, pad_classes=True).to(dev).train()
feat = torch.randn(150, 10, 1024, device=dev, requires_grad=True)
h0, c0 = torch.zeros(10, 33, device=dev), torch.zeros(10, 33, device=dev)
for _ in range(N):
    model.zero_grad()
    feat.grad = None
    model(feat, h0, c0).sum().backward()
# the losses' neighbours on the Charades-shaped batch
x, lab, Tb, L = (t.to(dev) for t in synth_noblank(0, 150, 256, 158, 20))
xb, y, Tbb, Lb = (t.to(dev) for t in synth_binary(0, 150, 256, 158, 20))
rows = (torch.rand(256, 20, 38, device=dev) < 0.08).int()
for _ in range(N):
    ctc_amd.noblank_best_path(x, lab, Tb, L)
    ctc_amd.noblank_posteriors(x, lab, Tb, L)
    ctc_amd.binary_best_path(xb, y, Tbb, Lb)
    ctc_amd.binary_posteriors(xb, y, Tbb, Lb)
    ctc_amd.dedup_multihot_targets(rows)
    xs = x.clone().requires_grad_(True)
    ctc_amd.noblank_ctc_loss(xs, lab, Tb, L, label_smoothing=0.9)[0].backward()
torch.cuda.synchronize()
print("done")
