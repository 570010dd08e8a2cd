"""The producer of the logits: drop-in for the reference's ``LSTM_cell`` (LSTM.py:21-51).

``LSTM_cell(args).forward(feat, v_hsn, v_csn)`` runs, per frame, ``v = self.v(feat[time])`` (Linear + BatchNorm1d +
ReLU + Dropout: torch's own layers, kept) and one ``nn.LSTMCell`` step whose hidden state is stored as
``v_series[time]`` -- the ``[T, B, C]`` tensor the CTC losses read.  Here the LSTMCell step and that store are ONE HIP
launch per frame (``ctc_amd_lstm_cell_step``, include/ctc_amd.h): gate products, cell update, and the hidden state
written straight into the logits tensor, optionally with one pad column (``pad_classes=True``: an odd class count such
as the reference's 33 gets rows of C + 1 floats, the extra logit -1e30 -- softmax gives it exactly 0, so no loss or
gradient value changes, and the rows become the even, 8-byte aligned rows of the fastest loss kernel).

Same attribute names as the reference (``v``, ``v.layers``, ``v_cell``): its checkpoints load unchanged.  The backward
pass through the cell is plain torch arithmetic on the gate activations the forward launch saved (the recurrence's
backward is not on the HIP path).  No CPU path: non-HIP tensors raise.
"""
import torch
import torch.nn as nn

from . import _lib
from . import functional as F

PAD_LOGIT = -1.0e30


class BasicModule(nn.Module):
    """feature head of one frame: Linear -> BatchNorm1d -> ReLU -> Dropout (LSTM.py:7-18), torch's own layers"""

    def __init__(self, inDim, outDim, dp_rate=0.3):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(inDim, outDim), nn.BatchNorm1d(outDim), nn.ReLU(), nn.Dropout(p=dp_rate))

    def forward(self, x):
        return self.layers(x)


def lstm_cell_step(x, h, c, w_ih, w_hh, b_ih, b_hh, series_row=None, pad_value=PAD_LOGIT, want_gates=False):
    """One fused LSTMCell step on the device -> (h', c', gates | None).  ``series_row``: a [B, cols >= H] view with unit
    stride over its last dimension (e.g. ``v_series[time]``) that receives h' in columns [0, H) and ``pad_value`` behind."""
    F._require_hip(x, "x")
    B, I = x.shape
    H = h.shape[1]
    dev = x.device
    args = [t if (t.dtype is torch.float32 and t.is_contiguous()) else t.float().contiguous()
            for t in (x, h, c, w_ih, w_hh, b_ih, b_hh)]
    h_out = torch.empty((B, H), dtype=torch.float32, device=dev)
    c_out = torch.empty((B, H), dtype=torch.float32, device=dev)
    gates = torch.empty((B, 4 * H), dtype=torch.float32, device=dev) if want_gates else None
    sp, ss, sc = None, 0, 0
    if series_row is not None:
        if series_row.dim() != 2 or series_row.shape[0] != B or series_row.shape[1] < H or series_row.stride(1) != 1 \
                or series_row.dtype is not torch.float32 or series_row.device != dev:
            raise ValueError("ctc_amd: series_row must be a float32 [B, >= H] view with unit stride over the classes")
        sp, ss, sc = series_row.data_ptr(), series_row.stride(0), series_row.shape[1]
    with F._on_device(dev):
        rc = _lib.load().ctc_amd_lstm_cell_step(*(t.data_ptr() for t in args), B, I, H, h_out.data_ptr(), c_out.data_ptr(),
                                                gates.data_ptr() if want_gates else None, sp, ss, sc, float(pad_value),
                                                F._stream_handle(dev))
    if rc:
        _lib.check(rc, "ctc_amd_lstm_cell_step")
    return h_out, c_out, gates


def lstm_series(v_all, h0, c0, w_ih, w_hh, b_ih, b_hh, cols=None, pad_value=PAD_LOGIT, want_backward_state=False):
    """All T LSTMCell steps in ONE launch (``ctc_amd_lstm_series``; the reference's class counts, I + H <= 80) ->
    (v_series [T,B,cols], gates [T,B,4H] | None, cells [T+1,B,H] | None), or None when the size is not one the launch
    takes (step frame by frame with ``lstm_cell_step`` then)."""
    F._require_hip(v_all, "v_all")
    T, B, I = v_all.shape
    H = h0.shape[1]
    cols = H if cols is None else int(cols)
    dev = v_all.device
    args = [t if (t.dtype is torch.float32 and t.is_contiguous()) else t.float().contiguous()
            for t in (v_all, h0, c0, w_ih, w_hh, b_ih, b_hh)]
    series = torch.empty((T, B, cols), dtype=torch.float32, device=dev)
    gates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev) if want_backward_state else None
    cells = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev) if want_backward_state else None
    with F._on_device(dev):
        rc = _lib.load().ctc_amd_lstm_series(*(t.data_ptr() for t in args), T, B, I, H, series.data_ptr(), series.stride(0),
                                             series.stride(1), cols, float(pad_value),
                                             gates.data_ptr() if want_backward_state else None,
                                             cells.data_ptr() if want_backward_state else None, None, None, F._stream_handle(dev))
    if rc == _lib.ERR_UNSUPPORTED_SHAPE:
        return None
    if rc:
        _lib.check(rc, "ctc_amd_lstm_series")
    return series, gates, cells


def lstm_series_backward(d_series, gates, cells, w_hh):
    """The backward recurrence of ``lstm_series`` in one launch (``ctc_amd_lstm_series_backward``) ->
    (dpre [T,B,4H], dh0 [B,H], dc0 [B,H]); ``d_series`` [T,B,>=H]: the upstream gradient of v_series."""
    F._require_hip(d_series, "d_series")
    T, B, G = gates.shape
    H = G // 4
    dev = gates.device
    ds = d_series if (d_series.dtype is torch.float32 and d_series.stride(2) == 1) else d_series.float().contiguous()
    w = w_hh if (w_hh.dtype is torch.float32 and w_hh.is_contiguous()) else w_hh.float().contiguous()
    dpre = torch.empty((T, B, G), dtype=torch.float32, device=dev)
    dh0 = torch.empty((B, H), dtype=torch.float32, device=dev)
    dc0 = torch.empty((B, H), dtype=torch.float32, device=dev)
    with F._on_device(dev):
        rc = _lib.load().ctc_amd_lstm_series_backward(ds.data_ptr(), ds.stride(0), ds.stride(1), gates.data_ptr(), cells.data_ptr(),
                                                      w.data_ptr(), T, B, H, dpre.data_ptr(), dh0.data_ptr(), dc0.data_ptr(),
                                                      F._stream_handle(dev))
    if rc:
        _lib.check(rc, "ctc_amd_lstm_series_backward")
    return dpre, dh0, dc0


class _SeriesFn(torch.autograd.Function):
    """v_all [T,B,I], (h0, c0), LSTMCell parameters -> v_series [T,B,cols]: one launch for the reference's class counts, T
    fused launches otherwise; backward = BPTT in torch."""

    @staticmethod
    def forward(ctx, v_all, h0, c0, w_ih, w_hh, b_ih, b_hh, cols, pad_value):
        T, B, _ = v_all.shape
        H = h0.shape[1]
        need = any(ctx.needs_input_grad[:7])
        ctx.H = H
        whole = lstm_series(v_all, h0, c0, w_ih, w_hh, b_ih, b_hh, cols, pad_value, want_backward_state=need)
        if whole is not None:
            series, gates, cells = whole
            ctx.one_launch = need
            if need:
                hs = torch.cat([h0.detach().float().unsqueeze(0), series[:, :, :H]])
                ctx.save_for_backward(v_all, w_ih, w_hh, hs, cells, gates)
            return series
        ctx.one_launch = False
        series = torch.empty((T, B, cols), dtype=torch.float32, device=v_all.device)
        hs, cs, gs = [h0], [c0], []
        h, c = h0, c0
        for t in range(T):
            h, c, g = lstm_cell_step(v_all[t], h, c, w_ih, w_hh, b_ih, b_hh, series[t], pad_value, want_gates=need)
            if need:
                hs.append(h); cs.append(c); gs.append(g)
        if need:
            ctx.save_for_backward(v_all, w_ih, w_hh, torch.stack(hs), torch.stack(cs), torch.stack(gs))
        return series

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_series):
        v_all, w_ih, w_hh, hs, cs, gs = ctx.saved_tensors
        T, H = v_all.shape[0], ctx.H
        if ctx.one_launch:                                   # the recurrence in one launch, the rest as GEMMs over all frames
            dpre, dh0, dc0 = lstm_series_backward(d_series, gs, cs, w_hh)
            flat = dpre.reshape(-1, 4 * H)
            db = flat.sum(0)
            return (dpre @ w_ih.float(), dh0, dc0, flat.t() @ v_all.reshape(-1, v_all.shape[2]).float(),
                    flat.t() @ hs[:-1].reshape(-1, H), db, db.clone(), None, None)
        dh = torch.zeros_like(hs[0])
        dc = torch.zeros_like(cs[0])
        dv = torch.empty_like(v_all)
        dw_ih, dw_hh = torch.zeros_like(w_ih), torch.zeros_like(w_hh)
        db = torch.zeros(4 * H, dtype=torch.float32, device=v_all.device)
        for t in range(T - 1, -1, -1):
            dh = dh + d_series[t, :, :H]
            i, f, g, o = gs[t].split(H, dim=1)
            tc = torch.tanh(cs[t + 1])
            dc = dc + dh * o * (1.0 - tc * tc)
            dpre = torch.cat([dc * g * i * (1.0 - i), dc * cs[t] * f * (1.0 - f), dc * i * (1.0 - g * g),
                              dh * tc * o * (1.0 - o)], dim=1)
            dv[t] = dpre @ w_ih
            dw_ih += dpre.t() @ v_all[t]
            dw_hh += dpre.t() @ hs[t]
            db += dpre.sum(0)
            dh = dpre @ w_hh
            dc = dc * f
        return dv, dh, dc, dw_ih, dw_hh, db, db.clone(), None, None


class LSTM_cell(nn.Module):
    """``LSTM_cell(args)`` as in the reference (``args.extract_feat_dim``, ``.v_class``, ``.batch_size``, ``.temporal``);
    ``forward(feat[T,B,feat_dim], v_hsn[B,C], v_csn[B,C]) -> v_series[T,B,C]`` (``[T,B,C+1]`` with ``pad_classes`` and odd C)."""

    def __init__(self, args, _BaseModule=BasicModule, pad_classes=False):
        super().__init__()
        self.args = args
        self.input_size = args.extract_feat_dim
        self.v_class = args.v_class
        self.batch_size = args.batch_size
        self.temporal = args.temporal
        self.pad_classes = bool(pad_classes)
        self.v = _BaseModule(self.input_size, self.v_class)
        self.v_cell = nn.LSTMCell(self.v_class, self.v_class)

    def forward(self, feat, v_hsn, v_csn):
        F._require_hip(feat, "feat")
        v_all = torch.stack([self.v(feat[time]) for time in range(self.temporal)])   # (BatchNorm statistics per frame, as the reference)
        H = self.v_class
        cols = H + 1 if (self.pad_classes and H % 2) else H
        cell = self.v_cell
        return _SeriesFn.apply(v_all, v_hsn, v_csn, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, cols, PAD_LOGIT)
