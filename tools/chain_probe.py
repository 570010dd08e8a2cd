#!/usr/bin/env python3
"""Diagnostic: the lattice chain of noblank_r16.hpp alone (all rows pre-published), cycles per step."""
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
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ctc_amd import _lib  # noqa: E402

lib = _lib.load()
fn = lib.ctc_amd_debug_chain_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
out = torch.zeros(16, dtype=torch.int64, device="cuda")
T, SP = 150, 20
MODES = {0: "others poll + sleep", 1: "others: VALU", 2: "others: LDS traffic", 3: "others: VALU + LDS (worker mix)",
         9: "others: VALU at priority 2", 11: "others: worker mix at priority 2"}
CASES = ((1, 1, 0), (16, 1, 0), (16, 256, 0), (16, 256, 1), (16, 256, 2), (16, 256, 3), (16, 256, 9), (16, 256, 11),
         (8, 256, 3), (12, 256, 3))
if os.environ.get("PROBE_QUICK"):
    CASES = ((1, 1, 0), (16, 256, 3))
for chain_b in ((2,) if os.environ.get("PROBE_QUICK") else (1, 2)):
    for waves, grid, mode in CASES:
        for _ in range(3):
            out.zero_()
            rc = fn(T, SP, waves, grid, out.data_ptr(), None, mode, chain_b)
            assert rc == 0, rc
            torch.cuda.synchronize()
        o = out.cpu().tolist()
        nloop = (T - 1) // 16 * 16
        print("chains on waves 0,%d  waves_alive=%2d grid=%3d %-34s alpha %6d cyc (%5.1f/step)  beta %6d cyc (%5.1f/step)   "
              "main loop only: %5.1f / %5.1f per step"
              % (chain_b, waves, grid, MODES[mode], o[0], o[0] / (T - 1), o[1], o[1] / (T - 1), o[8] / nloop, o[9] / nloop))
