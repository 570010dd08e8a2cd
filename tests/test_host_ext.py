"""The C++ autograd node of the eager path (ctc_amd/csrc/autograd_ext.cpp): it must issue the SAME launches as the Python
Function -- bit-identical loss, nll and gradient -- and leave everything it does not take to the Python Function."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import np_, synth_binary, synth_noblank


def test_host_extension_is_built_and_exports_the_two_calls():
    from ctc_amd import build
    so = build.build_host_ext()
    assert os.path.exists(so)
    import ctc_amd.functional as F
    ext = F._load_host_ext()
    assert ext is not None and callable(ext.ctc_loss) and callable(ext.set_abi)


def test_non_hip_arguments_take_the_python_function_and_raise_there():
    import ctc_amd
    import ctc_amd.functional as F
    x, lab, Tb, L = synth_noblank(0, 12, 3, 6, 4)
    assert F._fast_apply(x, lab, Tb, L, None) is None
    with pytest.raises(ctc_amd.CtcAmdError):
        ctc_amd.CTCLoss.apply(x, lab, Tb, L)


# ------------------------------------------------------------------ on the device
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    import ctc_amd.functional as F
    assert F._load_host_ext() is not None, "ctc_amd/lib/ext/ctc_amd_autograd_ext.so missing: run __graft_entry__.build()"
    return torch.device("cuda:0")


def _both(fn, args, dev, **kw):
    """fn through the C++ node and through the Python Function -> two result dicts"""
    import ctc_amd.functional as F
    out = []
    for use_ext in (True, False):
        saved = F._host_ext
        if not use_ext:
            F._host_ext = None
        try:
            x = args[0].to(dev).requires_grad_(True)
            rest = [a.to(dev) for a in args[1:]]
            res = fn(x, *rest, **kw)
            loss = res[0] if isinstance(res, tuple) else res
            node = loss.grad_fn.name()
            (loss * 0.5).backward()
            torch.cuda.synchronize()
            out.append({"loss": np_(loss), "nll": np_(res[1]) if isinstance(res, tuple) else None, "grad": np_(x.grad), "node": node})
        finally:
            F._host_ext = saved
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(150, 256, 158, 20), (40, 7, 33, 9), (300, 5, 38, 12)])
@pytest.mark.parametrize("int64", [False, True])
def test_noblank_cpp_node_is_bit_identical_to_the_python_function(dev, shape, int64):
    import ctc_amd
    T, B, C, S = shape
    args = synth_noblank(3, T, B, C, S, var_T=True, int64=int64)
    for fn in (ctc_amd.CTCLoss.apply, ctc_amd.noblank_ctc_loss):
        a, b = _both(fn, args, dev)
        assert "CtcFn" in a["node"] and "CtcFn" not in b["node"], (a["node"], b["node"])
        assert np.array_equal(a["loss"], b["loss"]) and np.array_equal(a["grad"], b["grad"])
        if a["nll"] is not None:
            assert np.array_equal(a["nll"], b["nll"])


@pytest.mark.gpu
def test_binary_cpp_node_is_bit_identical_and_honours_batch_total(dev):
    import ctc_amd
    args = synth_binary(5, 150, 64, 158, 20, var_T=True)
    a, b = _both(ctc_amd.binary_ctc_loss, args, dev, batch_total=512)
    assert "CtcFn" in a["node"] and "CtcFn" not in b["node"]
    for k in ("loss", "nll", "grad"):
        assert np.array_equal(a[k], b[k]), k
    c, _ = _both(ctc_amd.binary_ctc_loss, args, dev)
    assert abs(float(c["loss"]) / float(a["loss"]) - 8.0) < 1e-5


@pytest.mark.gpu
def test_cpp_node_retain_graph_inplace_loss_and_no_grad(dev):
    import ctc_amd
    x, lab, Tb, L = synth_noblank(7, 60, 16, 38, 10, var_T=True)
    xd = x.to(dev).requires_grad_(True)
    lab, Tb, L = lab.to(dev), Tb.to(dev), L.to(dev)
    loss = ctc_amd.CTCLoss.apply(xd, lab, Tb, L)
    assert "CtcFn" in loss.grad_fn.name()
    loss.backward(retain_graph=True)
    g1 = xd.grad.clone()
    xd.grad = None
    loss.backward()                                       # second backward: recomputed by the node
    assert torch.equal(g1, xd.grad)
    xd.grad = None
    loss2 = ctc_amd.CTCLoss.apply(xd, lab, Tb, L)
    loss2 /= 4                                            # a tensor of its own, not a view
    loss2.backward()
    assert torch.allclose(xd.grad * 4, g1, rtol=1e-6, atol=0)
    with torch.no_grad():
        l3 = ctc_amd.CTCLoss.apply(xd, lab, Tb, L)
    assert l3.grad_fn is None and not l3.requires_grad and torch.equal(l3, loss.detach())
    l4, nll4 = ctc_amd.noblank_ctc_loss(xd.detach(), lab, Tb, L)
    assert not l4.requires_grad and not nll4.requires_grad and torch.equal(l4, loss.detach())


@pytest.mark.gpu
def test_cpp_node_takes_strided_logits_and_leaves_the_rest_to_python(dev):
    import ctc_amd
    x, lab, Tb, L = synth_noblank(9, 50, 12, 40, 8)
    lab, Tb, L = lab.to(dev), Tb.to(dev), L.to(dev)
    big = torch.randn(50, 24, 40, device=dev)
    view = big[:, ::2].detach().requires_grad_(True)              # strides over T and B honoured, unit stride over C
    ref = view.detach().contiguous().requires_grad_(True)
    la = ctc_amd.CTCLoss.apply(view, lab, Tb, L)
    lb = ctc_amd.CTCLoss.apply(ref, lab, Tb, L)
    assert "CtcFn" in la.grad_fn.name()
    la.backward(); lb.backward()
    assert torch.equal(la, lb) and torch.equal(view.grad, ref.grad)
    # lengths on the host, a transposed logits view, python lists: the Python Function converts them
    xt = torch.randn(50, 40, 12, device=dev).transpose(1, 2).requires_grad_(True)      # [50,12,40], stride over C = 12
    l1 = ctc_amd.CTCLoss.apply(xt, lab, Tb.cpu(), L.cpu())
    assert "CtcFn" not in l1.grad_fn.name()
    l2 = ctc_amd.CTCLoss.apply(xt, lab, Tb, L)
    assert torch.equal(l1, l2)
    with pytest.raises(ValueError):
        ctc_amd.CTCLoss.apply(xt, lab[:5], Tb, L)


@pytest.mark.gpu
def test_cpp_node_is_graph_capturable_and_shares_the_python_workspace(dev):
    import ctc_amd
    import ctc_amd.functional as F
    x, lab, Tb, L = synth_noblank(11, 150, 256, 158, 20, var_T=True)
    xd = x.to(dev).requires_grad_(True)
    lab, Tb, L = lab.to(dev), Tb.to(dev), L.to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            xd.grad = None
            ctc_amd.CTCLoss.apply(xd, lab, Tb, L).backward()
        key = (dev.index, s.cuda_stream, ctc_amd._lib.NOBLANK)
        assert key in F._workspaces                      # the node ran on the Python layer's workspace of this stream
    torch.cuda.current_stream().wait_stream(s)
    eager_grad = xd.grad.clone()
    g = torch.cuda.CUDAGraph()
    xd.grad = None
    with torch.cuda.graph(g):
        loss = ctc_amd.CTCLoss.apply(xd, lab, Tb, L)
        loss.backward()
    xd.grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(xd.grad, eager_grad)
    assert ctc_amd.workspace_status() == 0


def test_host_extension_that_does_not_load_falls_back_quietly(monkeypatch, tmp_path):
    """The node is optional: an .so that cannot be imported (built against another torch, truncated) must make the first
    CTCLoss.apply fall back to the Python Function with ONE warning, not raise (round-3 advice)."""
    import warnings
    import ctc_amd.functional as F
    from ctc_amd import build
    bad = tmp_path / "ctc_amd_autograd_ext.so"
    bad.write_bytes(b"not an ELF file")
    monkeypatch.setattr(build, "HOST_EXT_SO", str(bad))
    monkeypatch.setattr(build, "host_ext_is_current", lambda: True)
    monkeypatch.setattr(F, "_host_ext", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert F._load_host_ext() is None
    assert len(w) == 1 and "Python Function" in str(w[0].message)
    assert F._host_ext is None
    # and a stamp for another torch version is "not current"
    monkeypatch.undo()
    monkeypatch.setattr(build, "HOST_EXT_STAMP", str(tmp_path / "stamp.txt"))
    (tmp_path / "stamp.txt").write_text("0.0.0")
    assert build.host_ext_is_current() is False


@pytest.mark.gpu
def test_cpp_node_is_once_differentiable_and_scales_on_the_current_stream(dev):
    """create_graph=True must raise like the Python Function's once_differentiable (the raw-kernel gradient carries no
    graph), and backward issued under another current stream scales the gradient on THAT stream (round-3 advice)."""
    import ctc_amd
    x, lab, Tb, L = synth_noblank(3, 40, 6, 20, 5)
    xd = x.to(dev).requires_grad_(True)
    args = (lab.to(dev), Tb.to(dev), L.to(dev))
    loss = ctc_amd.CTCLoss.apply(xd, *args)
    with pytest.raises(RuntimeError, match="differentiable once"):
        torch.autograd.grad(loss, xd, create_graph=True)
    # reference gradient (default stream), then the same with backward under a side stream and an upstream factor
    xd.grad = None
    ctc_amd.CTCLoss.apply(xd, *args).backward()
    torch.cuda.synchronize()
    ref = xd.grad.clone()
    xd.grad = None
    loss = ctc_amd.CTCLoss.apply(xd, *args)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        (3.0 * loss).backward()
    side.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(xd.grad, 3.0 * ref)
