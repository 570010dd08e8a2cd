"""the producer step of the logits -- the reference's LSTM_cell (LSTM.py:21-51) with the LSTMCell
step and the v_series[time] store as one HIP launch per frame (ctc_amd_lstm_cell_step), against the fixture captured
from the reference module itself, the numpy restatement, and torch.nn.LSTMCell on the device."""
import types

import numpy as np
import pytest
import torch

from oracle import ctc_numpy
from tests.helpers import np_, synth_noblank

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    import ctc_amd  # noqa: F401
    return torch.device("cuda:0")


def _args(feat=1024, C=33, B=10, T=10):
    return types.SimpleNamespace(extract_feat_dim=feat, v_class=C, batch_size=B, temporal=T)


def test_lstm_series_golden(dev, golden):
    """the fixture of the reference's own module (make_golden.py F7): cell loop on the device from the stored
    per-frame inputs, padded and unpadded rows"""
    from ctc_amd import producer
    f = golden("lstm_series")
    t = {k: torch.tensor(f[k]).to(dev) for k in ("v_in", "h0", "c0", "w_ih", "w_hh", "b_ih", "b_hh")}
    for cols in (33, 34, 40):
        series = producer._SeriesFn.apply(t["v_in"], t["h0"], t["c0"], t["w_ih"], t["w_hh"], t["b_ih"], t["b_hh"], cols,
                                          producer.PAD_LOGIT)
        torch.cuda.synchronize()
        got = np_(series)
        assert got.shape == (10, 10, cols)
        assert np.abs(got[:, :, :33] - f["v_series"]).max() < 5e-6
        assert (got[:, :, 33:] == np.float32(producer.PAD_LOGIT)).all()


@pytest.mark.parametrize("shape", [(1, 5, 3), (10, 33, 33), (37, 40, 17), (256, 158, 158), (300, 64, 200)])
def test_lstm_cell_step_vs_oracle(dev, shape):
    from ctc_amd import producer
    B, I, H = shape
    g = torch.Generator().manual_seed(B + I + H)
    x, h, c = (torch.randn(B, n, generator=g) for n in (I, H, H))
    w_ih, w_hh = 0.3 * torch.randn(4 * H, I, generator=g), 0.3 * torch.randn(4 * H, H, generator=g)
    b_ih, b_hh = torch.randn(4 * H, generator=g), torch.randn(4 * H, generator=g)
    ref_h, ref_c, ref_g = ctc_numpy.lstm_cell_step(*(np_(v) for v in (x, h, c, w_ih, w_hh, b_ih, b_hh)))
    series = torch.zeros(B, H + 3, device=dev)
    args = [v.to(dev) for v in (x, h, c, w_ih, w_hh, b_ih, b_hh)]
    hn, cn, gates = producer.lstm_cell_step(*args, series_row=series, pad_value=-7.0, want_gates=True)
    torch.cuda.synchronize()
    # (fp32 dot products of I + H terms against float64: 2e-5 on values of order 1)
    assert np.abs(np_(hn) - ref_h).max() < 2e-5 and np.abs(np_(cn) - ref_c).max() < 2e-5 * max(1.0, np.abs(ref_c).max())
    assert np.abs(np_(gates) - ref_g).max() < 2e-5
    assert (np_(series)[:, :H] == np_(hn)).all() and (np_(series)[:, H:] == -7.0).all()
    # torch's own cell on the device agrees as well (the arithmetic the reference calls)
    cell = torch.nn.LSTMCell(I, H).to(dev)
    with torch.no_grad():
        cell.weight_ih.copy_(args[3]); cell.weight_hh.copy_(args[4]); cell.bias_ih.copy_(args[5]); cell.bias_hh.copy_(args[6])
        th, tc = cell(args[0], (args[1], args[2]))
    assert (hn - th).abs().max().item() < 2e-5 and (cn - tc).abs().max().item() < 2e-5 * max(1.0, float(tc.abs().max()))


def test_lstm_cell_module_matches_torch_module_forward_and_backward(dev):
    """drop-in: same parameters (state_dict of a module built from torch layers the way LSTM.py builds it), same
    v_series in train-free (eval) mode, same gradients for the parameters and the features"""
    import torch.nn as nn
    from ctc_amd import producer
    a = _args(feat=64, C=33, B=6, T=7)
    torch.manual_seed(3)
    ours = producer.LSTM_cell(a).to(dev).eval()

    class Ref(nn.Module):                                    # LSTM.py:21-51 with torch's cell, on the test's device
        def __init__(self):
            super().__init__()
            self.v = producer.BasicModule(a.extract_feat_dim, a.v_class)
            self.v_cell = nn.LSTMCell(a.v_class, a.v_class)

        def forward(self, feat, h, c):
            out = []
            for time in range(a.temporal):
                h, c = self.v_cell(self.v(feat[time]), (h, c))
                out.append(h)
            return torch.stack(out)
    ref = Ref().to(dev).eval()
    ref.load_state_dict(ours.state_dict())                    # same names: v.layers.*, v_cell.*
    feat = torch.randn(a.temporal, a.batch_size, a.extract_feat_dim, device=dev)
    h0, c0 = 0.1 * torch.randn(a.batch_size, 33, device=dev), 0.1 * torch.randn(a.batch_size, 33, device=dev)
    f1, f2 = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    s1, s2 = ours(f1, h0, c0), ref(f2, h0, c0)
    assert (s1 - s2).abs().max().item() < 5e-6
    w = torch.randn_like(s2)
    (s1 * w).sum().backward()
    (s2 * w).sum().backward()
    assert (f1.grad - f2.grad).abs().max().item() < 2e-5
    for (n1, p1), (n2, p2) in zip(ours.named_parameters(), ref.named_parameters()):
        assert n1 == n2 and (p1.grad - p2.grad).abs().max().item() < 2e-4 * max(1.0, float(p2.grad.abs().max())), n1


def test_padded_series_feeds_the_fast_loss_kernel_unchanged(dev):
    """odd class count (the reference's 33): v_series with one pad column at -1e30 is an even-C input of the loss
    (the four-rows-per-wave kernel) with the SAME loss and the same gradient on the real classes, 0 on the pad"""
    import ctc_amd
    from ctc_amd import producer
    a = _args(feat=48, C=33, B=8, T=20)
    torch.manual_seed(5)
    m_pad = producer.LSTM_cell(a, pad_classes=True).to(dev).eval()
    m_raw = producer.LSTM_cell(a).to(dev).eval()
    m_raw.load_state_dict(m_pad.state_dict())
    feat = torch.randn(a.temporal, a.batch_size, a.extract_feat_dim, device=dev)
    h0, c0 = torch.zeros(a.batch_size, 33, device=dev), torch.zeros(a.batch_size, 33, device=dev)
    _, lab, Tb, L = synth_noblank(1, a.temporal, a.batch_size, 33, 6, var_T=True)
    out = {}
    for name, m in (("pad", m_pad), ("raw", m_raw)):
        m.zero_grad()
        series = m(feat, h0, c0)
        series.retain_grad()
        loss = ctc_amd.CTCLoss.apply(series, lab.to(dev), Tb.to(dev), L.to(dev))
        loss.backward()
        out[name] = (float(loss), np_(series.grad), np_(m.v_cell.weight_hh.grad))
    assert out["pad"][1].shape[2] == 34 and out["raw"][1].shape[2] == 33
    assert abs(out["pad"][0] - out["raw"][0]) < 1e-5 * max(1.0, abs(out["raw"][0]))
    assert np.abs(out["pad"][1][:, :, :33] - out["raw"][1]).max() < 1e-6 and np.abs(out["pad"][1][:, :, 33]).max() == 0.0
    assert np.abs(out["pad"][2] - out["raw"][2]).max() < 1e-5


@pytest.mark.parametrize("shape", [(150, 10, 33, 33), (20, 7, 38, 38), (12, 256, 33, 33), (9, 5, 17, 40), (3, 1, 64, 16)])
def test_lstm_series_one_launch_is_bit_identical_to_the_steps(dev, shape):
    """ctc_amd_lstm_series: the T frames in one launch -- same numbers, bit for bit, as T calls of ctc_amd_lstm_cell_step
    (v_series with and without pad columns, gate activations, cell states), and the numpy restatement within fp32."""
    import ctc_amd
    from ctc_amd import producer
    T, B, I, H = shape
    g = torch.Generator().manual_seed(sum(shape))
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)          # noqa: E731
    v_all, h0, c0 = rnd(T, B, I), rnd(B, H), rnd(B, H)
    w_ih, w_hh, b_ih, b_hh = rnd(4 * H, I) * 0.3, rnd(4 * H, H) * 0.3, rnd(4 * H) * 0.1, rnd(4 * H) * 0.1
    for cols in (H, H + 1 + (H % 2)):
        whole = ctc_amd.lstm_series(v_all, h0, c0, w_ih, w_hh, b_ih, b_hh, cols, want_backward_state=True)
        assert whole is not None, shape
        series, gates, cells = whole
        ref = torch.empty_like(series)
        h, c = h0, c0
        for t in range(T):
            h, c, gt = ctc_amd.lstm_cell_step(v_all[t], h, c, w_ih, w_hh, b_ih, b_hh, ref[t], want_gates=True)
            assert torch.equal(gt, gates[t]) and torch.equal(c, cells[t + 1]), (shape, t)
        torch.cuda.synchronize()
        assert torch.equal(series, ref), shape
        assert torch.equal(cells[0], c0)
    want = ctc_numpy.lstm_cell_series(np_(v_all), np_(h0), np_(c0), np_(w_ih), np_(w_hh), np_(b_ih), np_(b_hh))[0]
    assert np.abs(np_(series)[:, :, :H] - want).max() < 2e-5


def test_lstm_series_sizes_it_does_not_take_fall_back_to_the_steps(dev):
    """I + H > 80: the C ABI answers CTC_AMD_ERR_UNSUPPORTED_SHAPE, `lstm_series` returns None, the module steps frame by frame"""
    import ctc_amd
    from ctc_amd import producer
    T, B, H = 6, 4, 158
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)          # noqa: E731
    v_all, h0, c0 = rnd(T, B, H), rnd(B, H), rnd(B, H)
    w_ih, w_hh, b_ih, b_hh = rnd(4 * H, H) * 0.1, rnd(4 * H, H) * 0.1, rnd(4 * H) * 0.1, rnd(4 * H) * 0.1
    assert ctc_amd.lstm_series(v_all, h0, c0, w_ih, w_hh, b_ih, b_hh) is None
    series = producer._SeriesFn.apply(v_all, h0, c0, w_ih, w_hh, b_ih, b_hh, H, producer.PAD_LOGIT)
    want = ctc_numpy.lstm_cell_series(np_(v_all), np_(h0), np_(c0), np_(w_ih), np_(w_hh), np_(b_ih), np_(b_hh))[0]
    assert np.abs(np_(series) - want).max() < 2e-5


@pytest.mark.parametrize("shape", [(150, 10, 33, 33), (20, 7, 38, 38), (12, 64, 33, 33), (9, 5, 17, 40)])
def test_lstm_series_backward_one_launch_matches_torch_autograd(dev, shape):
    """ctc_amd_lstm_series_backward + the three GEMMs against torch.autograd through torch's own nn.LSTMCell loop on the
    device (same parameters, same inputs, a padded v_series and a random upstream gradient): every gradient."""
    from ctc_amd import producer
    T, B, I, H = shape
    g = torch.Generator().manual_seed(sum(shape))
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)          # noqa: E731
    cell = torch.nn.LSTMCell(I, H).to(dev)
    leaves = [rnd(T, B, I), rnd(B, H), rnd(B, H)]
    up = rnd(T, B, H + 1)
    out = {}
    for name in ("torch", "hip"):
        v_all, h0, c0 = (t.clone().requires_grad_(True) for t in leaves)
        cell.zero_grad()
        if name == "torch":
            h, c, rows = h0, c0, []
            for t in range(T):
                h, c = cell(v_all[t], (h, c))
                rows.append(h)
            series = torch.stack(rows)
            (series * up[:, :, :H]).sum().backward()
        else:
            series = producer._SeriesFn.apply(v_all, h0, c0, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, H + 1,
                                              producer.PAD_LOGIT)
            assert series.grad_fn.one_launch
            (series * up).sum().backward()
        out[name] = [np_(t.grad) for t in (v_all, h0, c0, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    names = ("dv", "dh0", "dc0", "dW_ih", "dW_hh", "db_ih", "db_hh")
    for a, b_, what in zip(out["hip"], out["torch"], names):
        scale = max(1.0, float(np.abs(b_).max()))
        assert np.abs(a - b_).max() <= 3e-5 * scale, (what, shape, np.abs(a - b_).max())
    # and against the float64 restatement of the same back-propagation (oracle/ctc_numpy.py)
    up64 = np_(up)[:, :, :H]
    want = ctc_numpy.lstm_cell_series_backward(up64, *(np_(t) for t in leaves), np_(cell.weight_ih), np_(cell.weight_hh),
                                               np_(cell.bias_ih), np_(cell.bias_hh))
    for a, b_, what in zip(out["hip"], want, names):
        scale = max(1.0, float(np.abs(b_).max()))
        assert np.abs(a - b_).max() <= 2e-5 * scale, (what, shape, np.abs(a - b_).max())


# ------------------------------------------------------------------ the head: Linear + BatchNorm1d + ReLU + Dropout (LSTM.py:8-18)
def _reference_args(temporal):
    import types
    return types.SimpleNamespace(extract_feat_dim=1024, v_class=33, batch_size=10, temporal=temporal)


def _load_head_state(model, d, dev):
    lin, bn = model.v.layers[0], model.v.layers[1]
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(d["lin_w"])); lin.bias.copy_(torch.from_numpy(d["lin_b"]))
        bn.weight.copy_(torch.from_numpy(d["bn_w"])); bn.bias.copy_(torch.from_numpy(d["bn_b"]))
        bn.running_mean.copy_(torch.from_numpy(d["rm0"])); bn.running_var.copy_(torch.from_numpy(d["rv0"]))
        bn.num_batches_tracked.zero_()
        c = model.v_cell
        c.weight_ih.copy_(torch.from_numpy(d["w_ih"])); c.weight_hh.copy_(torch.from_numpy(d["w_hh"]))
        c.bias_ih.copy_(torch.from_numpy(d["b_ih"])); c.bias_hh.copy_(torch.from_numpy(d["b_hh"]))
    return model.to(dev)


@pytest.mark.parametrize("tag", ["p0", "p3"])
def test_head_train_mode_golden(dev, golden, monkeypatch, tag):
    """The fused head against the reference's OWN LSTM_cell in TRAIN mode (tests/golden/lstm_head_train.npz, make_golden.py
    F8): BatchNorm on each frame's batch statistics, the dropout masks the reference drew (p = 0 and p = 0.3), the running
    statistics after six frames, and the reference's autograd gradient of every parameter and of the features."""
    import ctc_amd
    d = golden("lstm_head_train")
    T = d["feat"].shape[0]
    model = _load_head_state(ctc_amd.LSTM_cell(_reference_args(T)), d, dev).train()
    model.v.layers[3].p = 0.0 if tag == "p0" else 0.3
    mask = torch.from_numpy(d["mask_" + tag]).to(dev)
    if tag == "p3":                                          # the mask the reference drew stands in for torch's draw
        monkeypatch.setattr(torch.nn.functional, "dropout", lambda x, p, training: mask)
    feat = torch.from_numpy(d["feat"]).to(dev).requires_grad_(True)
    h0, c0, R = (torch.from_numpy(d[k]).to(dev) for k in ("h0", "c0", "R"))
    # the head alone, through the C ABI
    lin, bn = model.v.layers[0], model.v.layers[1]
    out, _lin, mean, var, inv = ctc_amd.head_forward(feat.detach(), lin.weight, lin.bias, bn.weight, bn.bias, eps=bn.eps,
                                                     mask=mask if tag == "p3" else None, want_backward_state=True)
    assert np.abs(np_(out) - d["head_" + tag]).max() < 2e-5
    # the module: v_series, running statistics, every gradient
    v_series = model(feat, h0, c0)
    assert np.abs(np_(v_series) - d["v_series_" + tag]).max() < 1e-5
    assert np.abs(np_(bn.running_mean) - d["rm_" + tag]).max() < 1e-5 and np.abs(np_(bn.running_var) - d["rv_" + tag]).max() < 1e-5
    assert int(bn.num_batches_tracked) == int(d["nbt_" + tag]) == T
    (v_series * R).sum().backward()
    got = {"d_feat": feat.grad, "d_lin_w": lin.weight.grad, "d_lin_b": lin.bias.grad, "d_bn_w": bn.weight.grad,
           "d_bn_b": bn.bias.grad, "d_w_ih": model.v_cell.weight_ih.grad}
    for k, g in got.items():
        want = d[k + "_" + tag]
        # (d_lin_b is rounding noise on both sides: batch statistics remove a per-column constant)
        scale = max(0.05, float(np.abs(want).max()))
        assert np.abs(np_(g) - want).max() <= 2e-4 * scale, (k, tag, np.abs(np_(g) - want).max(), scale)


@pytest.mark.parametrize("shape", [(5, 10, 1024, 33), (3, 256, 1024, 158), (4, 37, 64, 40), (2, 2, 16, 5)])
@pytest.mark.parametrize("train", [True, False])
def test_head_matches_torch_layers_forward_and_backward(dev, shape, train):
    """ctc_amd_head_forward + its backward against torch's own nn.Linear / nn.BatchNorm1d / nn.ReLU applied frame by frame
    on the device (train: batch statistics and the running-statistics update; eval: running statistics), with a dropout mask."""
    from ctc_amd import producer
    T, B, K, C = shape
    g = torch.Generator().manual_seed(sum(shape) + int(train))
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)          # noqa: E731
    lin, bn = torch.nn.Linear(K, C).to(dev), torch.nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(rnd(C) * 0.5 + 1.0); bn.bias.copy_(rnd(C) * 0.2)
        bn.running_mean.copy_(rnd(C) * 0.3); bn.running_var.copy_(rnd(C) * 0.4 + 1.0)
    bn.train(train)
    mask = ((rnd(T, B, C) > -0.4).float() / 0.7)
    up = rnd(T, B, C)
    leaves = rnd(T, B, K)
    res = {}
    for name in ("torch", "hip"):
        feat = leaves.clone().requires_grad_(True)
        lin.zero_grad(); bn.zero_grad()
        rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
        if name == "torch":
            out = torch.stack([torch.relu(bn(lin(feat[t]))) for t in range(T)]) * mask
        else:
            if train:
                out, mean, var = producer._HeadFn.apply(feat, lin.weight, lin.bias, bn.weight, bn.bias, None, None, bn.eps, mask)
                assert mean.shape == (T, C) and not mean.requires_grad
            else:
                out = producer._HeadFn.apply(feat, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, mask)[0]
        (out * up).sum().backward()
        res[name] = [np_(out)] + [np_(t.grad) for t in (feat, lin.weight, lin.bias, bn.weight, bn.bias)]
        if name == "torch":                                  # (the layers updated their running statistics: put them back)
            with torch.no_grad():
                bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
    for a, b_, what in zip(res["hip"], res["torch"], ("out", "d_feat", "d_W", "d_b", "d_gamma", "d_beta")):
        if train and what == "d_b":                          # identically 0 under batch statistics: rounding noise on both sides,
            continue                                         # amplified by 1 / sqrt(var) when a column's two rows nearly agree
        scale = max(1.0, float(np.abs(b_).max()))
        assert np.abs(a - b_).max() <= 1e-4 * scale, (what, shape, train, np.abs(a - b_).max(), scale)


def test_head_shapes_the_launch_does_not_take_fall_back_to_the_layers(dev):
    """B > 256 or a feature dimension that is not a multiple of 16: CTC_AMD_ERR_UNSUPPORTED_SHAPE -> head_forward returns None
    and LSTM_cell applies self.v frame by frame (as it does for any head that is not the reference's BasicModule)."""
    import ctc_amd
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)          # noqa: E731
    w, b_, gm, bt = rnd(8, 24), rnd(8), rnd(8), rnd(8)
    assert ctc_amd.head_forward(rnd(2, 4, 24), w, b_, gm, bt) is None                   # K % 16
    assert ctc_amd.head_forward(rnd(2, 300, 32), rnd(8, 32), b_, gm, bt) is None        # B > 256
    import types
    args = types.SimpleNamespace(extract_feat_dim=24, v_class=8, batch_size=4, temporal=3)
    model = ctc_amd.LSTM_cell(args).to(dev).eval()
    ref = torch.stack([model.v(x) for x in rnd(3, 4, 24)])
    assert ref.shape == (3, 4, 8)
    out = model(rnd(3, 4, 24), rnd(4, 8), rnd(4, 8))
    assert out.shape == (3, 4, 8) and bool(torch.isfinite(out).all())
