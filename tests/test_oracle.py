"""The oracle against the golden vectors captured from the reference (CPU only)."""
import numpy as np
import pytest

from oracle import ctc_c, ctc_numpy

NOBLANK = ["kat1_noblank", "kat3_noblank", "cfg1_noblank", "charades_noblank", "edge_noblank"]
BINARY = ["kat2_binary", "cfg1_binary", "charades_binary", "edge_binary"]
BLANK = ["blank_small", "blank_long"]


def _check(r, d, nll_tol, grad_tol):
    # float32 forward keeps the reference's operation order: nll agrees to an ulp or two
    assert np.abs(r["nll"] - d["nll"]).max() <= nll_tol * max(1.0, np.abs(d["nll"]).max())
    assert abs(float(r["loss"]) - float(d["loss"])) <= nll_tol * max(1.0, abs(float(d["loss"])))
    assert np.abs(r["grad"] - d["grad"]).max() <= grad_tol


@pytest.mark.parametrize("impl", [ctc_numpy, ctc_c], ids=["numpy", "c"])
@pytest.mark.parametrize("name", NOBLANK)
def test_noblank_vs_reference(golden, impl, name):
    d = golden(name)
    _check(impl.noblank_ctc(d["x"], d["lab"], d["in_len"], d["tgt_len"], np.float32), d, 3e-7, 5e-5)
    _check(impl.noblank_ctc(d["x"], d["lab"], d["in_len"], d["tgt_len"], np.float64), d, 1e-6, 1e-5)


@pytest.mark.parametrize("impl", [ctc_numpy, ctc_c], ids=["numpy", "c"])
@pytest.mark.parametrize("name", BINARY)
def test_binary_vs_reference(golden, impl, name):
    d = golden(name)
    _check(impl.binary_ctc(d["x"], d["y"], d["in_len"], d["tgt_len"], np.float32), d, 1e-6, 1e-6)
    _check(impl.binary_ctc(d["x"], d["y"], d["in_len"], d["tgt_len"], np.float64), d, 2e-6, 1e-6)


@pytest.mark.parametrize("impl", [ctc_numpy, ctc_c], ids=["numpy", "c"])
@pytest.mark.parametrize("name", BLANK)
def test_blank_vs_torch(golden, impl, name):
    d = golden(name)
    _check(impl.blank_ctc(d["lp"], d["tgt"], d["in_len"], d["tgt_len"], np.float32), d, 1e-6, 1e-5)
    _check(impl.blank_ctc(d["lp"], d["tgt"], d["in_len"], d["tgt_len"], np.float64), d, 5e-6, 1e-5)


def test_known_answers(golden):
    # SURVEY section 4 KAT table (values produced by the shipped reference modules)
    d = golden("kat1_noblank")
    assert abs(float(d["loss"]) - 6.62364197) < 1e-6
    assert np.allclose(d["nll"], [7.22587299, 6.02141142], atol=1e-6)
    assert np.allclose(d["grad"][0, 0], [0.06152686, 0.18483686, -0.42485094, 0.01123994, 0.16724733], atol=1e-7)
    assert abs(float(golden("kat2_binary")["loss"]) - 4.01820946) < 1e-6
    assert abs(float(golden("kat3_noblank")["loss"]) - 7.22587299) < 1e-6


def test_blank_matches_torch_live():
    torch = pytest.importorskip("torch")
    from tests.helpers import synth_blank
    lp, tgt, Tb, L = synth_blank(11, 60, 5, 17, 9, var_T=True)
    lp = lp.requires_grad_(True)
    loss = torch.nn.functional.ctc_loss(lp, tgt, Tb, L, blank=0, reduction="mean")
    loss.backward()
    r = ctc_c.blank_ctc(lp.detach().numpy(), tgt.numpy(), Tb.numpy(), L.numpy())
    assert abs(float(r["loss"]) - float(loss)) < 1e-5
    assert np.abs(r["grad"] - lp.grad.numpy()).max() < 1e-5


@pytest.mark.parametrize("which", ["noblank", "binary"])
def test_gradient_is_derivative_of_loss(which):
    """closed-form alpha-beta gradient == finite differences of the float64 loss."""
    rng = np.random.default_rng(3)
    T, B, C, S = 7, 3, 5, 4
    x = rng.standard_normal((T, B, C))
    L = np.array([4, 2, 1]); Tb = np.array([7, 5, 7])
    if which == "noblank":
        tg = rng.integers(0, C, (B, S)); tg[0, 1] = tg[0, 0]
        f = lambda z: ctc_numpy.noblank_ctc(z, tg, Tb, L, np.float64)
    else:
        tg = rng.random((B, S, C))
        f = lambda z: ctc_numpy.binary_ctc(z, tg, Tb, L, np.float64)
    g = f(x)["grad"]
    num = np.zeros_like(x)
    for i in np.ndindex(*x.shape):
        xp = x.copy(); xp[i] += 1e-6
        xm = x.copy(); xm[i] -= 1e-6
        num[i] = (f(xp)["loss"] - f(xm)["loss"]) / 2e-6
    assert np.abs(num - g).max() < 1e-7
    assert np.abs(g[6, 1]).max() == 0.0          # rows t >= T_b carry no gradient


def test_numpy_and_c_agree_on_random_shapes():
    from tests.helpers import synth_noblank, synth_binary, np_
    for seed, (T, B, C, S) in enumerate([(20, 4, 10, 5), (33, 6, 70, 64), (10, 10, 33, 10)]):
        x, lab, Tb, L = map(np_, synth_noblank(seed, T, B, C, S, var_T=True))
        a = ctc_numpy.noblank_ctc(x, lab, Tb, L, np.float64)
        c = ctc_c.noblank_ctc(x, lab, Tb, L, np.float64, threads=2)
        assert np.abs(a["nll"] - c["nll"]).max() < 1e-9 and np.abs(a["grad"] - c["grad"]).max() < 1e-12
        x, y, Tb, L = map(np_, synth_binary(seed, T, B, C, S, var_T=True, density=0.2))
        a = ctc_numpy.binary_ctc(x, y, Tb, L, np.float64)
        c = ctc_c.binary_ctc(x, y, Tb, L, np.float64, threads=2)
        assert np.abs(a["nll"] - c["nll"]).max() < 1e-9 and np.abs(a["grad"] - c["grad"]).max() < 1e-12


def test_label_smoothing_restatement_gradient():
    """NoBlankCTC.py:100-107 is a comment in the reference (no behaviour to capture): the restatement's
    closed-form gradient is checked against central differences, and lambda = 1 must be the plain loss."""
    rng = np.random.default_rng(3)
    T, B, C, S = 8, 3, 6, 4
    x = rng.normal(size=(T, B, C))
    lab = rng.integers(0, C, (B, S))
    il, tl = np.array([8, 6, 8]), np.array([4, 2, 3])
    plain = ctc_numpy.noblank_ctc(x, lab, il, tl, np.float64)
    one = ctc_numpy.noblank_ctc(x, lab, il, tl, np.float64, label_smoothing=1.0)
    assert np.allclose(one["nll"], plain["nll"], atol=1e-12) and np.allclose(one["grad"], plain["grad"], atol=1e-12)
    r = ctc_numpy.noblank_ctc(x, lab, il, tl, np.float64, label_smoothing=0.9)
    assert (r["nll"] < plain["nll"] + 10).all() and abs(r["loss"] - plain["loss"]) > 1e-3
    eps, g = 1e-6, np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        g[idx] = (ctc_numpy.noblank_ctc(xp, lab, il, tl, np.float64, want_grad=False, label_smoothing=0.9)["loss"] -
                  ctc_numpy.noblank_ctc(xm, lab, il, tl, np.float64, want_grad=False, label_smoothing=0.9)["loss"]) / (2 * eps)
    assert np.abs(g - r["grad"]).max() < 1e-7
    assert np.abs(r["grad"].sum(axis=2)).max() < 1e-12          # rows still sum to zero
    assert np.abs(r["grad"][6:, 1]).max() == 0.0                # rows beyond T_b


def _reference_dedup_ops(rows_b):
    """charades_ctc_next_pred.py:646-651,654-678 typed with the reference's own tensor operations (IntTensor
    codes, Python 2**o, `not in`) -- the same typing tests/golden/make_golden.py F6 captured the fixture with."""
    import torch
    S, C = rows_b.shape
    tgt = torch.tensor(rows_b, dtype=torch.int32)
    code = torch.IntTensor(S).zero_()
    for t in range(S):
        for o in range(C):
            code[t] += tgt[t, o] * 2 ** o
    only, only_code, n = torch.IntTensor(S, C).zero_(), torch.IntTensor(S).zero_(), 0
    for t in range(S):
        if code[t] not in only_code:
            only_code[t] = code[t]
            only[n] = tgt[t]
            n += 1
    for pad in range(S - n):
        only[n + pad] = -1
    return only.numpy(), n


@pytest.mark.parametrize("S,C", [(6, 5), (12, 9), (20, 30), (8, 31), (8, 32), (8, 33), (10, 38), (6, 64)])
def test_dedup_targets_restatement_matches_the_reference_algorithm_in_torch(S, C):
    """Pins the numpy restatement on the reference's tensor operations at every class count up to 64, the
    reference's defaults (opts.py:60-61: 38 / 33) included: int32 codes wrap -- class 31 is the sign bit,
    classes >= 32 drop out, code 0 never enters."""
    rng = np.random.default_rng(5 + C)
    rows = (rng.random((6, S, C)) < 0.15).astype(np.int32)
    rows[0, 3] = rows[0, 1]                               # a repeat that is not adjacent
    rows[1, 2] = 0                                        # an empty row in the middle
    rows[2] = rows[2, :1]                                 # one distinct row only
    rows[3] = 0                                           # nothing at all
    if C > 33:
        rows[4] = 0
        rows[4, 0, [1, 33]] = 1                           # rows that differ only in classes >= 32 ...
        rows[4, 1, [1, C - 1]] = 1
        rows[4, 2, C - 2] = 1                             # ... and rows made only of them
        rows[4, 3, 2] = 1
    rows[5, :, min(31, C - 1)] = 1                        # the sign bit (C >= 32) in every row
    got, length = ctc_numpy.dedup_multihot_targets(rows)
    for b in range(6):
        only, n = _reference_dedup_ops(rows[b])
        assert n == int(length[b]) and (only == got[b]).all(), (b, n, int(length[b]))
    assert int(length[3]) == 0 and int(length[2]) <= 1
    if C > 33:
        assert int(length[4]) == 2                        # {1,33} and {2}: the reference's own answer at C = 38


def test_dedup_targets_golden(golden):
    """the fixture make_golden.py F6 captured from the reference's tensor operations (torch version inside)"""
    f = golden("dedup_targets")
    assert int(f["overflow_at_65"]) == 1
    for C in (5, 30, 31, 32, 33, 38, 64):
        got, length = ctc_numpy.dedup_multihot_targets(f["rows_%d" % C])
        assert (length == f["len_%d" % C]).all() and (got == f["out_%d" % C]).all(), C
    assert f["len_38"].tolist()[1] == 2                   # rows {1,33}, {1,35}, {36}, {2}, {1} -> {1,33}, {2}
    # exact-row comparison (not the reference) keeps what the int32 code merges
    _, exact = ctc_numpy.dedup_multihot_targets(f["rows_38"], exact_rows=True)
    assert int(exact[1]) == 5 and int(exact[2]) == 3


def test_dedup_targets_more_than_64_classes_raise_like_the_reference():
    rows = np.zeros((1, 3, 65), np.int32)
    with pytest.raises(OverflowError):
        _reference_dedup_ops(rows[0])
    with pytest.raises(OverflowError):
        ctc_numpy.dedup_multihot_targets(rows)
    out, n = ctc_numpy.dedup_multihot_targets(rows, exact_rows=True)
    assert int(n[0]) == 0


def test_lstm_series_restatement_matches_the_reference_module(golden):
    """LSTM.py:39-51 (the reference's own LSTM_cell, run in eval mode by make_golden.py F7): the numpy restatement of
    the LSTMCell loop reproduces v_series and the final cell state from the stored per-frame inputs and parameters"""
    f = golden("lstm_series")
    for dt, tol in ((np.float64, 2e-6), (np.float32, 5e-6)):
        series, h, c = ctc_numpy.lstm_cell_series(f["v_in"], f["h0"], f["c0"], f["w_ih"], f["w_hh"], f["b_ih"], f["b_hh"], dt)
        assert series.shape == f["v_series"].shape == (10, 10, 33)
        assert np.abs(series - f["v_series"]).max() < tol and np.abs(c - f["c_final"]).max() < tol
        assert np.abs(h - f["v_series"][-1]).max() < tol


def test_lstm_series_backward_restatement_matches_torch_autograd():
    """the BPTT restatement (oracle/ctc_numpy.py) against torch.autograd through torch.nn.LSTMCell on the CPU, float64"""
    import torch
    T, B, I, H = 12, 5, 7, 6
    g = torch.Generator().manual_seed(5)
    cell = torch.nn.LSTMCell(I, H).double()
    v = torch.randn(T, B, I, generator=g, dtype=torch.float64, requires_grad=True)
    h0 = torch.randn(B, H, generator=g, dtype=torch.float64, requires_grad=True)
    c0 = torch.randn(B, H, generator=g, dtype=torch.float64, requires_grad=True)
    up = torch.randn(T, B, H, generator=g, dtype=torch.float64)
    h, c, rows = h0, c0, []
    for t in range(T):
        h, c = cell(v[t], (h, c))
        rows.append(h)
    (torch.stack(rows) * up).sum().backward()
    want = [t.grad.numpy() for t in (v, h0, c0, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    p = [t.detach().numpy() for t in (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    got = ctc_numpy.lstm_cell_series_backward(up.numpy(), v.detach().numpy(), h0.detach().numpy(), c0.detach().numpy(), *p)
    for a, b in zip(got, want):
        assert np.abs(a - b).max() < 1e-12
