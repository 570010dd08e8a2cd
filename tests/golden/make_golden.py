#!/usr/bin/env python3
"""Capture golden input/output vectors from the *reference* loss modules.

Runs ONLY in the authoring container (needs /root/reference, which never
travels to the GPU box).  It imports the reference's two loss files
unmodified -- `NoBlankCTC.py` and `NoBlankBinaryCTC.py` -- with `.cuda()`
made an identity (the container has no GPU; the files hard-code `.cuda()`,
NoBlankCTC.py:40-96), runs forward + autograd backward on seeded inputs and
stores inputs AND outputs as small .npz fixtures next to this script.

The fixtures are data (inputs and expected outputs); no reference source text
is stored.  Blank-CTC fixtures (F5) come from torch.nn.functional.ctc_loss on
CPU, the third-party arithmetic behind models/layers/AsyncTFCriterion.py:198.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py F1 F2      # a subset
"""
import os
import sys
import time

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CTC_REFERENCE", "/root/reference")


def _import_reference():
    # .cuda() -> identity, tensors and modules alike (CPU-only container)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from NoBlankCTC import NoBlankCTC  # noqa
    from NoBlankBinaryCTC import NoBlankBinaryCTC  # noqa
    return NoBlankCTC, NoBlankBinaryCTC


def run_noblank(mod_cls, x, lab, in_len, tgt_len):
    """-> (mean loss, per-sample nll, grad wrt x) from the reference module."""
    xt = torch.tensor(x, dtype=torch.float32, requires_grad=True)
    labt = torch.tensor(lab)
    il = torch.tensor(in_len, dtype=torch.int64)
    tl = torch.tensor(tgt_len, dtype=torch.int64)
    m = mod_cls()
    loss = m(xt, labt, il, tl)
    loss.backward()
    # per-sample nll: rerun each sample alone (mean over a batch of one)
    nll = []
    for b in range(x.shape[1]):
        with torch.no_grad():
            lb = mod_cls()(torch.tensor(x[:, b:b + 1]), labt[b:b + 1], il[b:b + 1], tl[b:b + 1])
        nll.append(float(lb))
    return (np.float32(loss.item()), np.asarray(nll, np.float32),
            xt.grad.detach().numpy().astype(np.float32))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def synth_noblank(seed, T, B, C, S, var_T=True, int64=False):
    """SURVEY 8(d) generator: randn logits, L in [1,S], -1 padded labels."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, C, generator=g)
    L = torch.randint(1, S + 1, (B,), generator=g)
    lab = torch.randint(0, C, (B, S), generator=g, dtype=torch.int32)
    for b in range(B):
        lab[b, int(L[b]):] = -1
    if var_T:
        lo = min(S, T)
        Tb = torch.randint(lo, T + 1, (B,), generator=g)
        Tb = torch.maximum(Tb, L)
    else:
        Tb = torch.full((B,), T, dtype=torch.int64)
    if int64:
        lab = lab.long()
    return x.numpy(), lab.numpy(), Tb.numpy().astype(np.int64), L.numpy().astype(np.int64)


def synth_binary(seed, T, B, C, S, var_T=True, density=0.05):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, C, generator=g)
    L = torch.randint(1, S + 1, (B,), generator=g)
    y = (torch.rand(B, S, C, generator=g) < density).float()
    for b in range(B):
        y[b, int(L[b]):] = 0.0   # pad rows 0: nn.BCELoss rejects -1
    if var_T:
        Tb = torch.maximum(torch.randint(min(S, T), T + 1, (B,), generator=g), L)
    else:
        Tb = torch.full((B,), T, dtype=torch.int64)
    return x.numpy(), y.numpy(), Tb.numpy().astype(np.int64), L.numpy().astype(np.int64)


# ---------------------------------------------------------------- fixtures
def F1(NB, NBB):
    """KAT-1/2/3: inputs embedded in the reference's test.py (:258-269, :384-397)."""
    x2 = np.array([[[1.2, 2.3, 1.4, -0.5, 2.2], [-0.1, 1.2, 0.4, 2.5, 3.2]],
                   [[0.5, 1.3, 2.2, 0.1, 2.4], [1.1, 2.2, 0.7, 1.4, 2.2]],
                   [[0.8, -1.5, 2.3, 1.2, 2.1], [0.9, 1.4, 0.6, 2.3, 1.0]],
                   [[0.2, -1.0, 1.3, 2.2, 0.1], [0.2, 1.0, 1.6, 1.3, 1.2]]], np.float32)
    lab2 = np.array([[2, 3, 4], [1, 2, 0]], np.int64)
    loss, nll, grad = run_noblank(NB, x2, lab2, [4, 4], [3, 2])
    save("kat1_noblank", x=x2, lab=lab2, in_len=np.array([4, 4], np.int64),
         tgt_len=np.array([3, 2], np.int64), loss=loss, nll=nll, grad=grad)
    x1 = x2[:, :1].copy()
    y1 = np.array([[[0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1]]], np.float32)
    loss, nll, grad = run_noblank(NBB, x1, y1, [4], [3])
    save("kat2_binary", x=x1, y=y1, in_len=np.array([4], np.int64),
         tgt_len=np.array([3], np.int64), loss=loss, nll=nll, grad=grad)
    loss, nll, grad = run_noblank(NB, x1, lab2[:1], [4], [3])
    save("kat3_noblank", x=x1, lab=lab2[:1], in_len=np.array([4], np.int64),
         tgt_len=np.array([3], np.int64), loss=loss, nll=nll, grad=grad)


def F2(NB, NBB):
    """config 1: B=4 T=20 C=10 S=5, variable T_b and L_b, -1 padded int32 labels."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(20, 4, 10, generator=g).numpy()
    L = np.array([5, 3, 1, 4], np.int64)
    Tb = np.array([20, 17, 20, 9], np.int64)
    lab = torch.randint(0, 10, (4, 5), generator=g, dtype=torch.int32).numpy()
    for b in range(4):
        lab[b, L[b]:] = -1
    loss, nll, grad = run_noblank(NB, x, lab, Tb, L)
    save("cfg1_noblank", x=x, lab=lab, in_len=Tb, tgt_len=L, loss=loss, nll=nll, grad=grad)
    xb, y, Tb2, L2 = synth_binary(1, 20, 4, 10, 5, density=0.3)
    loss, nll, grad = run_noblank(NBB, xb, y, Tb2, L2)
    save("cfg1_binary", x=xb, y=y, in_len=Tb2, tgt_len=L2, loss=loss, nll=nll, grad=grad)


def F3(NB, NBB):
    """Charades-shaped reduced batch: B=8 T=150 C=158 S=20 for both variants (~20-40 s each)."""
    x, lab, Tb, L = synth_noblank(2, 150, 8, 158, 20, var_T=True)
    t0 = time.time()
    loss, nll, grad = run_noblank(NB, x, lab, Tb, L)
    print("  noblank T=150 B=8: %.1f s" % (time.time() - t0))
    save("charades_noblank", x=x, lab=lab, in_len=Tb, tgt_len=L, loss=loss, nll=nll, grad=grad)
    xb, y, Tb2, L2 = synth_binary(3, 150, 8, 158, 20, var_T=True)
    t0 = time.time()
    loss, nll, grad = run_noblank(NBB, xb, y, Tb2, L2)
    print("  binary T=150 B=8: %.1f s" % (time.time() - t0))
    save("charades_binary", x=xb, y=y, in_len=Tb2, tgt_len=L2, loss=loss, nll=nll, grad=grad)


def F4(NB, NBB):
    """edge set: repeated labels, L=1, L=S, T_b=L_b (single path), T_b<T, int64 labels."""
    T, B, C, S = 12, 8, 7, 6
    g = torch.Generator().manual_seed(4)
    x = (2.0 * torch.randn(T, B, C, generator=g)).numpy()
    lab = np.array([[0, 0, 2, 2, 2, 5],      # repeated consecutive labels, L=S
                    [3, -1, -1, -1, -1, -1],  # L=1
                    [6, 5, 4, 3, 2, 1],      # L=S, T_b=L_b: single feasible path
                    [1, 2, 1, 2, -1, -1],    # alternating repeats
                    [4, 4, 4, 4, 4, 4],      # all the same class
                    [0, 1, -1, -1, -1, -1],
                    [2, 6, 0, -1, -1, -1],
                    [5, 5, 1, 0, 0, -1]], np.int64)
    L = np.array([6, 1, 6, 4, 6, 2, 3, 5], np.int64)
    Tb = np.array([12, 12, 6, 7, 12, 2, 3, 11], np.int64)
    loss, nll, grad = run_noblank(NB, x, lab, Tb, L)
    save("edge_noblank", x=x, lab=lab, in_len=Tb, tgt_len=L, loss=loss, nll=nll, grad=grad)
    # binary edge: soft (non 0/1) targets, dense rows, empty rows, larger |x| (<=15)
    gb = torch.Generator().manual_seed(5)
    xb = (4.0 * torch.randn(T, 5, C, generator=gb)).clamp(-14, 14).numpy()
    y = torch.rand(5, S, C, generator=gb)
    y[0] = (y[0] < 0.5).float()
    y[1] = 0.0
    y[2] = 1.0
    y = y.numpy()
    Lb = np.array([6, 2, 6, 1, 4], np.int64)
    Tbb = np.array([12, 5, 6, 12, 9], np.int64)
    for b in range(5):
        y[b, Lb[b]:] = 0.0
    loss, nll, grad = run_noblank(NBB, xb, y, Tbb, Lb)
    save("edge_binary", x=xb, y=y, in_len=Tbb, tgt_len=Lb, loss=loss, nll=nll, grad=grad)


def F5(NB, NBB):
    """blank-CTC: torch.nn.functional.ctc_loss(blank=0, reduction='mean'), CPU."""
    import torch.nn.functional as Fn
    for name, (T, B, C, S, seed) in {"blank_small": (50, 4, 20, 8, 6),
                                     "blank_long": (400, 2, 100, 30, 7)}.items():
        g = torch.Generator().manual_seed(seed)
        lp = torch.randn(T, B, C, generator=g).log_softmax(2).requires_grad_(True)
        tgt = torch.randint(1, C, (B, S), generator=g)
        L = torch.randint(1, S + 1, (B,), generator=g)
        Tb = torch.randint(2 * S + 1, T + 1, (B,), generator=g)
        if name == "blank_small":
            tgt[0, 1] = tgt[0, 0]; tgt[0, 2] = tgt[0, 0]   # repeats force blanks
            L[0] = S
            Tb[0] = T
            L[1] = 1
        loss = Fn.ctc_loss(lp, tgt, Tb, L, blank=0, reduction="mean", zero_infinity=False)
        loss.backward()
        nll = Fn.ctc_loss(lp.detach(), tgt, Tb, L, blank=0, reduction="none")
        save(name, lp=lp.detach().numpy(), tgt=tgt.numpy(), in_len=Tb.numpy(), tgt_len=L.numpy(),
             loss=np.float32(loss.item()), nll=nll.numpy(), grad=lp.grad.numpy())


def reference_dedup_ops(rows_b):
    """One clip through the tensor operations of datasets/charades_ctc_next_pred.py:646-651 (row code) and
    :654-678 (walk, -1 padding), typed here on a given [S, C] block of label rows: the dataset file itself
    cannot be imported (torchvision, the Charades corpus), and the dedup sits in the middle of a 400-line
    method.  IntTensor codes, Python `2**o`, `not in` -- whatever this torch does with them is the answer."""
    S, C = rows_b.shape
    o_target = torch.tensor(rows_b, dtype=torch.int32)
    o_target_10 = torch.IntTensor(S).zero_()
    for t in range(S):
        for o in range(C):
            o_target_10[t] += o_target[t, o] * 2 ** o
    o_only_target = torch.IntTensor(S, C).zero_()
    o_only_target_10 = torch.IntTensor(S).zero_()
    o_target_length = 0
    for t in range(S):
        if o_target_10[t] not in o_only_target_10:
            o_only_target_10[t] = o_target_10[t]
            o_only_target[o_target_length] = o_target[t]
            o_target_length += 1
    if o_target_length < S:
        for pad in range(S - o_target_length):
            o_only_target[o_target_length + pad] = -1
    return o_only_target.numpy().copy(), o_target_length


def dedup_cases(C, S=10, seed=11):
    """[B, S, C] int32 label rows exercising the int32 row code at class count C."""
    rng = np.random.default_rng(seed + C)
    B = 7
    rows = np.zeros((B, S, C), np.int32)
    rows[0] = rng.random((S, C)) < 0.15
    rows[0, 5] = rows[0, 1]                                   # a repeat that is not adjacent
    rows[0, 3] = 0                                            # an empty row in the middle

    def hot(b, t, *cls):
        for c in cls:
            if c < C:
                rows[b, t, c] = 1
    hot(1, 0, 1, 33); hot(1, 1, 1, 35); hot(1, 2, 36); hot(1, 3, 2); hot(1, 4, 1)       # differ only at >= 32
    hot(2, 0, 32); hot(2, 1, 33, 37); hot(2, 2, 63); hot(2, 4, 0, 63)                    # only classes >= 32
    hot(3, 0, 31); hot(3, 1, 31, 0); hot(3, 2, 30); hot(3, 3, 31, 32); hot(3, 4, 30, 31)  # the sign bit
    rows[5, 0, 0] = 2; rows[5, 1, 1] = 1                      # values other than 0 / 1: 2 * 2**0 == 1 * 2**1
    if C > 30:
        rows[5, 2, 30] = 2                                    # 2 * 2**30 wraps to the sign bit ...
        hot(5, 3, 31)                                         # ... which is what class 31 codes to
        rows[5, 4, 30] = 4                                    # 4 * 2**30 wraps to 0: never enters
    rows[5, 5, 0] = -1                                        # a negative entry (the -1 padding fed back in)
    rows[6] = rng.random((S, C)) < 0.5                        # dense rows
    rows[6, 7:] = rows[6, :3]
    return rows                                               # rows[4] stays empty: a clip without any label


def F6(NB, NBB):
    """target dedup at class counts around the int32 wrap, the reference's defaults (38 object /
    33 verb classes, opts.py:60-61) included."""
    arrs = {"torch_version": np.array(torch.__version__)}
    for C in (5, 30, 31, 32, 33, 38, 64):
        rows = dedup_cases(C)
        out = np.zeros_like(rows)
        length = np.zeros(rows.shape[0], np.int64)
        for b in range(rows.shape[0]):
            out[b], length[b] = reference_dedup_ops(rows[b])
        arrs["rows_%d" % C], arrs["out_%d" % C], arrs["len_%d" % C] = rows, out, length
        print("  C=%d lengths %s" % (C, length.tolist()))
    try:
        reference_dedup_ops(np.zeros((2, 65), np.int32))
        arrs["overflow_at_65"] = np.array(0)
    except OverflowError as e:
        print("  C=65: OverflowError(%s)" % e)
        arrs["overflow_at_65"] = np.array(1)
    save("dedup_targets", **arrs)


def F7(NB, NBB):
    """the producer of the logits: the reference's OWN LSTM_cell (LSTM.py, imported unmodified, `.cuda()`
    an identity) in eval mode (BatchNorm on its running statistics, Dropout off: deterministic) at its production
    sizes -- extract_feat_dim 1024, v_class 33, batch 10, temporal 10 (opts.py:28,61,65; ctc_exe.py:14).  Stored: the
    per-frame inputs of the LSTMCell (self.v(feat[time])), its parameters, the initial state and v_series."""
    import types
    from LSTM import LSTM_cell
    torch.manual_seed(7)
    args = types.SimpleNamespace(extract_feat_dim=1024, v_class=33, batch_size=10, temporal=10)
    model = LSTM_cell(args).eval()
    with torch.no_grad():
        model.v.layers[1].running_mean.normal_(0.0, 0.3)      # (BatchNorm statistics of a trained head, not 0 / 1)
        model.v.layers[1].running_var.uniform_(0.5, 1.5)
        feat = torch.randn(args.temporal, args.batch_size, 1024)
        h0, c0 = 0.1 * torch.randn(args.batch_size, 33), 0.1 * torch.randn(args.batch_size, 33)
        v_series = model(feat, h0, c0)
        v_in = torch.stack([model.v(feat[t]) for t in range(args.temporal)])
        # the final state: the module returns only v_series; redo the loop with the module's own cell
        h, c = h0, c0
        for t in range(args.temporal):
            h, c = model.v_cell(v_in[t], (h, c))
    assert torch.equal(h, v_series[-1])
    cell = model.v_cell
    save("lstm_series", v_in=v_in.numpy(), h0=h0.numpy(), c0=c0.numpy(), w_ih=cell.weight_ih.detach().numpy(),
         w_hh=cell.weight_hh.detach().numpy(), b_ih=cell.bias_ih.detach().numpy(), b_hh=cell.bias_hh.detach().numpy(),
         v_series=v_series.numpy(), c_final=c.numpy(), torch_version=np.array(torch.__version__))


def F8(NB, NBB):
    """the head of the producer in TRAIN mode (LSTM.py:8-18,46-50): the reference's own LSTM_cell, BatchNorm on
    the statistics of each frame's batch (and its running statistics updated frame after frame), Dropout p = 0 and p = 0.3
    with the masks it drew captured by a forward hook.  Stored: inputs, parameters, masks, the head's per-frame outputs,
    v_series, the running statistics afterwards, and the reference's autograd gradients of sum(v_series * R)."""
    import types
    from LSTM import LSTM_cell
    torch.manual_seed(8)
    args = types.SimpleNamespace(extract_feat_dim=1024, v_class=33, batch_size=10, temporal=6)
    model = LSTM_cell(args).train()
    lin, bn, drop = model.v.layers[0], model.v.layers[1], model.v.layers[3]
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0.0, 0.2)
        bn.running_mean.normal_(0.0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    feat = torch.randn(args.temporal, args.batch_size, 1024)
    h0, c0 = 0.1 * torch.randn(args.batch_size, 33), 0.1 * torch.randn(args.batch_size, 33)
    R = torch.randn(args.temporal, args.batch_size, 33)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    arrs = dict(feat=feat.numpy(), h0=h0.numpy(), c0=c0.numpy(), R=R.numpy(), lin_w=lin.weight.detach().numpy(),
                lin_b=lin.bias.detach().numpy(), bn_w=bn.weight.detach().numpy(), bn_b=bn.bias.detach().numpy(),
                rm0=rm0.numpy(), rv0=rv0.numpy(), eps=np.array(bn.eps), momentum=np.array(bn.momentum),
                torch_version=np.array(torch.__version__))
    cell = model.v_cell
    arrs.update(w_ih=cell.weight_ih.detach().numpy(), w_hh=cell.weight_hh.detach().numpy(),
                b_ih=cell.bias_ih.detach().numpy(), b_hh=cell.bias_hh.detach().numpy())
    for tag, p in (("p0", 0.0), ("p3", 0.3)):
        drop.p = p
        with torch.no_grad():
            bn.running_mean.copy_(rm0)
            bn.running_var.copy_(rv0)
            bn.num_batches_tracked.zero_()
        masks, heads = [], []

        def hook(_m, inp, out):
            x = inp[0].detach()
            masks.append(torch.where(x > 0, out.detach() / x.clamp_min(1e-30), torch.zeros_like(x)))
            heads.append(out.detach().clone())
        hd = drop.register_forward_hook(hook)
        torch.manual_seed(80)
        f = feat.clone().requires_grad_(True)
        for prm in model.parameters():
            prm.grad = None
        v_series = model(f, h0, c0)
        (v_series * R).sum().backward()
        hd.remove()
        arrs.update({"mask_" + tag: torch.stack(masks).numpy(), "head_" + tag: torch.stack(heads).numpy(),
                     "v_series_" + tag: v_series.detach().numpy(), "rm_" + tag: bn.running_mean.numpy().copy(),
                     "rv_" + tag: bn.running_var.numpy().copy(), "nbt_" + tag: np.array(int(bn.num_batches_tracked)),
                     "d_feat_" + tag: f.grad.numpy(), "d_lin_w_" + tag: lin.weight.grad.numpy().copy(),
                     "d_lin_b_" + tag: lin.bias.grad.numpy().copy(), "d_bn_w_" + tag: bn.weight.grad.numpy().copy(),
                     "d_bn_b_" + tag: bn.bias.grad.numpy().copy(), "d_w_ih_" + tag: cell.weight_ih.grad.numpy().copy()})
        print("  %s: kept %.3f of the activations, |v_series| max %.3f" % (tag, float((torch.stack(masks) > 0).float().mean()),
                                                                           float(v_series.abs().max())))
    save("lstm_head_train", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(1)
    if sys.argv[1:] == ["F6"]:                                # needs torch only, not the reference's modules
        F6(None, None)
        sys.exit(0)
    NB, NBB = _import_reference()
    which = sys.argv[1:] or ["F1", "F2", "F3", "F4", "F5", "F6", "F7", "F8"]
    for w in which:
        print(w)
        globals()[w](NB, NBB)
