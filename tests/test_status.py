"""A bounded in-launch wait that runs out must be LOUD: NaN in the outputs it could not produce and a bit
in the workspace status word (include/ctc_amd.h, ctc_amd_workspace_status).  The hand-offs never break in
the product library, so this builds a fault-injection variant (-DCTC_AMD_FAULT_INJECT: sample 0's alpha
chain and sample 1's gradient workers of the no-blank kernel, a tile wave of sample 2 of the streamed binary
kernel pretend their wait ran out) and drives it in a subprocess."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import numpy as np, torch, ctc_amd
    from tests.helpers import synth_noblank, np_
    dev = torch.device("cuda:0")
    x, lab, Tb, L = synth_noblank(0, 150, 4, 158, 20)
    xd = x.to(dev).requires_grad_(True)
    loss, nll = ctc_amd.noblank_ctc_loss(xd, lab.to(dev), Tb.to(dev), L.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    nll, g = np_(nll), np_(xd.grad)
    assert np.isnan(nll[0]) and np.isfinite(nll[1:]).all(), nll        # the starved chain: NaN, not a number
    assert np.isnan(float(loss))
    assert np.isnan(g[:, 1]).any() and np.isfinite(g[:, 2:]).all()     # the starved workers' rows
    st = ctc_amd.workspace_status(clear=False)
    assert st == 1, st
    try:
        ctc_amd.check_status()
        raise SystemExit("check_status did not raise")
    except ctc_amd.CtcAmdError as e:
        assert "status 1" in str(e)
    assert ctc_amd.workspace_status() == 0                             # cleared by check_status
    # the streamed binary kernel: a tile wave of sample 2 reports a wait that ran out -> that sample's loss and every one
    # of its gradient rows are NaN, the others untouched, status bit 2
    from tests.helpers import synth_binary
    x, y, Tb, L = synth_binary(0, 150, 5, 158, 20)
    xd = x.to(dev).requires_grad_(True)
    loss, nll = ctc_amd.binary_ctc_loss(xd, y.to(dev), Tb.to(dev), L.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    nll, g = np_(nll), np_(xd.grad)
    assert np.isnan(nll[2]) and np.isfinite(np.delete(nll, 2)).all(), nll
    assert np.isnan(g[:, 2]).all() and np.isfinite(np.delete(g, 2, axis=1)).all()
    assert ctc_amd.workspace_status() == 2
    print("FAULT-INJECTION-OK")
""")


@pytest.mark.gpu
def test_starved_wait_is_loud():
    from ctc_amd import build
    so = build.build_fault()              # prebuilt by __graft_entry__.build(); compiled here only if missing / stale
    env = dict(os.environ, CTC_AMD_LIB=so, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FAULT-INJECTION-OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_product_library_reports_clean_status():
    import torch
    import ctc_amd
    from tests.helpers import synth_blank, synth_noblank
    dev = torch.device("cuda:0")
    x, lab, Tb, L = synth_noblank(1, 150, 8, 158, 20)
    ctc_amd.noblank_ctc_loss(x.to(dev).requires_grad_(True), lab.to(dev), Tb.to(dev), L.to(dev))[0].backward()
    lp, tgt, Tb, L = synth_blank(2, 300, 4, 40, 20)
    ctc_amd.set_blank_schedule(1)
    try:
        ctc_amd.blank_ctc_loss(lp.to(dev).requires_grad_(True), tgt.to(dev), Tb.to(dev), L.to(dev))[0].backward()
    finally:
        ctc_amd.set_blank_schedule(-1)
    assert ctc_amd.workspace_status() == 0
    ctc_amd.check_status()
