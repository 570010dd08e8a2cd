"""The producer of the logits: drop-in for the reference's ``LSTM_cell`` (LSTM.py:21-51).

``LSTM_cell(args).forward(feat, v_hsn, v_csn)`` runs, per frame, ``v = self.v(feat[time])`` (Linear + BatchNorm1d +
ReLU + Dropout) and one ``nn.LSTMCell`` step whose hidden state is stored as ``v_series[time]`` -- the ``[T, B, C]``
tensor the CTC losses read.  Here the head of ALL frames is one HIP launch (``ctc_amd_head_forward``: the 1024 -> C
product on the matrix cores, BatchNorm with the statistics of each frame's batch in train mode / the running statistics
in eval mode, ReLU, and the dropout mask torch drew), and the LSTMCell steps and stores are one more
(``ctc_amd_lstm_series``; one launch per frame, ``ctc_amd_lstm_cell_step``, at other sizes): gate products, cell update, and the hidden state
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


def head_forward(feat, weight, bias, bn_weight, bn_bias, running_mean=None, running_var=None, eps=1e-5, mask=None,
                 want_backward_state=False):
    """``dropout(relu(batchnorm(feat[t] @ weight.T + bias)))`` for every frame t in ONE launch (``ctc_amd_head_forward``)
    -> (out [T,B,C], lin [T,B,C] | None, mean [T,C] | None, var [T,C] | None, invstd [T,C] | None), or None when the launch
    does not take the shape (B > 256, K not a multiple of 16, unaligned rows).  running_mean / running_var: eval mode;
    both None: the statistics of each frame's batch.  ``mask``: [T,B,C], already scaled by 1 / (1 - p)."""
    F._require_hip(feat, "feat")
    T, B, K = feat.shape
    C = weight.shape[0]
    dev = feat.device
    f = feat if (feat.dtype is torch.float32 and feat.stride(2) == 1) else feat.float().contiguous()
    prm = [t if (t.dtype is torch.float32 and t.is_contiguous()) else t.float().contiguous()
           for t in (weight, bias, bn_weight, bn_bias)]
    rm = rv = None
    if running_mean is not None:
        rm, rv = running_mean.float().contiguous(), running_var.float().contiguous()
    mk = None if mask is None else (mask if (mask.dtype is torch.float32 and mask.is_contiguous()) else mask.float().contiguous())
    out = torch.empty((T, B, C), dtype=torch.float32, device=dev)
    train = rm is None
    lin = torch.empty((T, B, C), dtype=torch.float32, device=dev) if want_backward_state else None
    mean = torch.empty((T, C), dtype=torch.float32, device=dev) if train else None
    var = torch.empty((T, C), dtype=torch.float32, device=dev) if train else None
    inv = torch.empty((T, C), dtype=torch.float32, device=dev) if train else None
    ptr = lambda t: None if t is None else t.data_ptr()
    with F._on_device(dev):
        rc = _lib.load().ctc_amd_head_forward(f.data_ptr(), f.stride(0), f.stride(1), *(t.data_ptr() for t in prm), ptr(rm), ptr(rv),
                                              float(eps), ptr(mk), T, B, K, C, out.data_ptr(), out.stride(0), out.stride(1),
                                              ptr(lin), ptr(mean), ptr(var), ptr(inv), F._stream_handle(dev))
    if rc == _lib.ERR_UNSUPPORTED_SHAPE:
        return None
    if rc:
        _lib.check(rc, "ctc_amd_head_forward")
    return out, lin, mean, var, inv


class _HeadFn(torch.autograd.Function):
    """feat [T,B,K], Linear and BatchNorm parameters, running statistics (eval) or None (train), mask -> the head's output
    [T,B,C] (+ the batch statistics, non-differentiable).  Forward: one HIP launch.  Backward: elementwise torch arithmetic on
    the saved Linear output and statistics (BatchNorm's own formulas) and two GEMMs over all frames at once."""

    @staticmethod
    def forward(ctx, feat, weight, bias, bn_weight, bn_bias, running_mean, running_var, eps, mask):
        need = any(ctx.needs_input_grad[:5])
        res = head_forward(feat, weight, bias, bn_weight, bn_bias, running_mean, running_var, eps, mask, want_backward_state=need)
        if res is None:
            raise _lib.CtcAmdError("ctc_amd: the fused head does not take this shape (B <= 256, feature dimension a multiple of 16)")
        out, lin, mean, var, inv = res
        ctx.train = running_mean is None
        ctx.eps = float(eps)
        if need:
            if not ctx.train:
                mean, inv = running_mean.float().unsqueeze(0), torch.rsqrt(running_var.float() + float(eps)).unsqueeze(0)
            ctx.save_for_backward(feat, weight, bn_weight, bn_bias, lin, mean, inv, mask)
        stats = (mean, var) if ctx.train else (None, None)
        if ctx.train:
            ctx.mark_non_differentiable(mean, var)
        return (out,) + stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, _dm=None, _dv=None):
        feat, weight, gamma, beta, lin, mean, inv, mask = ctx.saved_tensors
        T, B, C = lin.shape
        xhat = (lin - mean.unsqueeze(1)) * inv.unsqueeze(1)
        dy = d_out.float()
        if mask is not None:
            dy = dy * mask
        dy = dy * ((xhat * gamma + beta) > 0)
        dbeta = dy.sum((0, 1))
        dgamma = (dy * xhat).sum((0, 1))
        dxh = dy * gamma
        if ctx.train:                                        # BatchNorm over the B rows of each frame
            dlin = inv.unsqueeze(1) * (dxh - dxh.mean(1, keepdim=True) - xhat * (dxh * xhat).mean(1, keepdim=True))
        else:
            dlin = dxh * inv.unsqueeze(1)
        flat = dlin.reshape(T * B, C)
        dW = flat.t() @ feat.reshape(T * B, -1).float()
        dfeat = (flat @ weight.float()).reshape(feat.shape) if ctx.needs_input_grad[0] else None
        return dfeat, dW, flat.sum(0), dgamma, dbeta, None, None, None, None


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

    def _head(self, feat):
        """self.v applied to every frame (LSTM.py:48): one launch when self.v is the reference's BasicModule and the shape
        is one the launch takes, the module itself frame by frame otherwise (a custom _BaseModule, B > 256, ...)."""
        T = self.temporal
        layers = getattr(self.v, "layers", None)
        std = (isinstance(layers, nn.Sequential) and len(layers) == 4 and isinstance(layers[0], nn.Linear)
               and isinstance(layers[1], nn.BatchNorm1d) and isinstance(layers[2], nn.ReLU) and isinstance(layers[3], nn.Dropout)
               and layers[0].bias is not None and layers[1].affine and layers[1].track_running_stats)
        B, K = feat.shape[1], feat.shape[2]
        if not std or feat.shape[0] < T or B > 256 or K % 16 or (self.training and (B < 2 or layers[1].momentum is None)):
            return torch.stack([self.v(feat[time]) for time in range(T)])
        lin, bn, drop = layers[0], layers[1], layers[3]
        x = feat[:T]
        mask = None
        if self.training and drop.p > 0:                     # torch draws the mask (its Philox stream), one call for all frames
            mask = torch.nn.functional.dropout(torch.ones((T, B, lin.out_features), dtype=torch.float32, device=feat.device),
                                               drop.p, True)
        if not self.training:
            return _HeadFn.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, mask)[0]
        out, mean, var = _HeadFn.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, None, None, bn.eps, mask)
        with torch.no_grad():                                # the running statistics, as T per-frame calls would leave them
            m = float(bn.momentum)
            coef = m * (1.0 - m) ** torch.arange(T - 1, -1, -1, dtype=torch.float32, device=feat.device)
            keep = (1.0 - m) ** T
            bn.running_mean.mul_(keep).add_(coef @ mean)
            bn.running_var.mul_(keep).add_(coef @ (var * (B / (B - 1.0))))
            bn.num_batches_tracked += T
        return out

    def forward(self, feat, v_hsn, v_csn):
        F._require_hip(feat, "feat")
        v_all = self._head(feat)                             # (BatchNorm statistics per frame, as the reference)
        H = self.v_class
        cols = H + 1 if (self.pad_classes and H % 2) else H
        cell = self.v_cell
        return _SeriesFn.apply(v_all, v_hsn, v_csn, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, cols, PAD_LOGIT)
