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
