#!/usr/bin/env python3
"""Diagnostic: timeline of the fused blank-CTC launch (CTC_AMD_BLANK_DEBUG must include 128).
Slots: 0 worker 0 enters; 1+k alpha chain of sample 0 at step 256 k; 10 that chain done;
11 worker 0 done gathering; 12/13/14 worker 0 / last / middle worker done.  Not part of the product path."""
import os
import subprocess
import sys

ROOT_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT_)
if "CTC_AMD_LIB" not in os.environ:                           # diagnostics live in the -DCTC_AMD_DIAGNOSTICS build only
    _diag = os.path.join(ROOT_, "ctc_amd", "lib", "libctc_amd_diag.so")   # (set BEFORE ctc_amd is imported)
    if not os.path.exists(_diag):
        subprocess.check_call([sys.executable, "-m", "ctc_amd.build", "--diag"], cwd=ROOT_, stdout=subprocess.DEVNULL)
    os.environ["CTC_AMD_LIB"] = _diag
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

wl = bench.Workload("blank", 64, 64, torch.device("cuda:0"), 0)
ws = wl.new_workspace()
loss = torch.zeros(4, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    wl.fused(loss.data_ptr(), ws, s)
torch.cuda.synchronize()
ws[64:256].zero_()
wl.fused(loss.data_ptr(), ws, s)
torch.cuda.synchronize()
st = ws[:256].cpu().numpy()[64:256].view(np.uint64).astype(np.int64)
if int(os.environ.get("CTC_AMD_BLANK_DEBUG", "0")) & 256:     # the chain probe: cycles per group of 16 steps, by what they went to
    names = ["emission rows there? (flag polls, waits for loaders)", "ring reads (16 x ds_read_b128, waited for)",
             "16 steps: arithmetic + store issue", "consumed flag + store-landing wait (s_waitcnt vmcnt)", "progress publication"]
    for h, what in enumerate(("first half (worker pool idle)", "second half (beside the streaming workers)")):
        n = max(int(st[6 * h + 5]), 1)
        tot = sum(int(st[6 * h + k]) for k in range(5))
        print("  alpha chain of sample 0, %s: %d groups of 16 steps, %.0f cycles per group = %.1f per step" % (what, n, tot / n, tot / n / 16))
        for k in range(5):
            print("      %-58s %7.0f cycles per group (%4.1f %%)" % (names[k], st[6 * h + k] / n, 100.0 * st[6 * h + k] / max(tot, 1)))
    sys.exit(0)
t0 = min(v for v in st[:16] if v > 0)
print("  alpha chain of sample 0: %d waits for emission rows, %d shader clocks in them" % (st[17], st[16]))
for i, v in enumerate(st[:16]):
    if v > 0:
        print("  slot %2d: %8.2f us" % (i, (v - t0) / 100.0))
