#!/usr/bin/env python3
"""Diagnostic: per-phase cycle stamps of workgroup 0 of the fused no-blank kernel.

    CTC_AMD_DEBUG_STOP=-1 python tools/stamps.py     # wave 0 (alpha chain wave)
    CTC_AMD_DEBUG_STOP=-3 python tools/stamps.py     # wave 2 (a rows-only wave)
Prints shader-clock cycles and 100-MHz realtime ticks between phase boundaries and the
clock they imply.  Not part of the product path.
"""
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

_B = int(sys.argv[2]) if len(sys.argv) > 2 else 256          # (more than #CUs: the persistent form -- the LAST sample's stamps remain)
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "noblank", _B, _B, torch.device("cuda:0"), 0)
ws = wl.new_workspace()
loss = torch.zeros(4, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    wl.fused(loss.data_ptr(), ws, s)
torch.cuda.synchronize()
ws[64:].zero_()
wl.fused(loss.data_ptr(), ws, s)
torch.cuda.synchronize()
st = ws.cpu().numpy()[64:64 + 12 * 16].view(np.uint64).reshape(12, 2).astype(np.int64)
if os.environ.get("CTC_AMD_DEBUG_STOP") == "-50":
    names = ["first wg entry", "first wg exit", "middle wg entry", "middle wg exit", "last wg entry", "last wg exit"]
    names += ["first wg alpha wave done", "middle wg alpha wave done", "last wg alpha wave done"]
    for i in range(9):
        if st[i][1] > 0:
            print("  %-26s %7.2f us (realtime, relative to the first workgroup's entry)" % (names[i], (st[i][1] - st[0][1]) / 100.0))
    sys.exit(0)
print("slot: us since entry (slots mean different things per kernel/wave, see the kernel source)")
for i in range(12):
    if st[i][1] > 0:
        print("  slot %d: %7.2f us  %8d cyc" % (i, (st[i][1] - st[0][1]) / 100.0, st[i][0] - st[0][0]))

