"""HIP path vs the oracle / golden vectors, through the C ABI.  Needs an MI355X.

Tolerances (BASELINE north_star: loss and input-gradient within 1e-4 of the reference,
fp32): per-sample nll and the mean loss relative 1e-5 (abs 1e-4 on values of a few
hundred), gradients abs 1e-4 against the reference's own autograd gradient and abs
2e-6*max(1, 256/B) against the float64 oracle at full size.
"""
import numpy as np
import pytest
import torch

from oracle import ctc_c, ctc_numpy
from tests.helpers import np_, synth_binary, synth_blank, synth_noblank

pytestmark = pytest.mark.gpu

NLL_RTOL = 1e-5
GRAD_ATOL_REF = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    import ctc_amd  # noqa: F401  (raises if libctc_amd.so is missing)
    return torch.device("cuda:0")


def _schedule(monkeypatch, mode):
    """blank-CTC schedule for this test only: 1 / 0 force / forbid the persistent launch, -1 the library's choice"""
    import ctc_amd
    ctc_amd.set_blank_schedule(mode)
    request_undo.append(lambda: ctc_amd.set_blank_schedule(-1))


request_undo = []


@pytest.fixture(autouse=True)
def _restore_schedule():
    yield
    while request_undo:
        request_undo.pop()()


def run_hip(fn, x, tg, il, tl, dev, lens_on_gpu=True, **kw):
    xd = torch.as_tensor(x, dtype=torch.float32).to(dev).requires_grad_(True)
    tgd = torch.as_tensor(tg).to(dev)
    ild, tld = torch.as_tensor(il), torch.as_tensor(tl)
    if lens_on_gpu:
        ild, tld = ild.to(dev), tld.to(dev)
    loss, nll = fn(xd, tgd, ild, tld, **kw)
    loss.backward()
    torch.cuda.synchronize()
    return {"loss": float(loss.detach()), "nll": np_(nll), "grad": np_(xd.grad)}


def assert_close(r, ref, grad_atol, nll_rtol=NLL_RTOL):
    scale = np.maximum(1.0, np.abs(ref["nll"]))
    assert (np.abs(r["nll"] - ref["nll"]) <= nll_rtol * scale).all(), np.abs(r["nll"] - ref["nll"]).max()
    assert abs(r["loss"] - float(ref["loss"])) <= nll_rtol * max(1.0, abs(float(ref["loss"])))
    assert np.isfinite(r["grad"]).all()
    err = np.abs(r["grad"] - ref["grad"]).max()
    assert err <= grad_atol, err


# ------------------------------------------------------------------ no-blank
@pytest.mark.parametrize("name", ["kat1_noblank", "kat3_noblank", "cfg1_noblank", "charades_noblank",
                                  "edge_noblank"])
def test_noblank_golden(golden, dev, name):
    import ctc_amd
    d = golden(name)
    r = run_hip(ctc_amd.noblank_ctc_loss, d["x"], d["lab"], d["in_len"], d["tgt_len"], dev)
    assert_close(r, d, GRAD_ATOL_REF)
    # rows beyond T_b carry exactly zero gradient
    for b, tb in enumerate(d["in_len"]):
        assert np.abs(r["grad"][int(tb):, b]).max(initial=0.0) == 0.0


def test_noblank_modules_and_function_surface(golden, dev):
    import ctc_amd
    d = golden("cfg1_noblank")
    x = torch.tensor(d["x"]).to(dev).requires_grad_(True)
    lab = torch.tensor(d["lab"]).to(dev)          # int32, -1 padded, as the dataset emits
    il, tl = torch.tensor(d["in_len"]).to(dev), torch.tensor(d["tgt_len"]).to(dev)
    loss = ctc_amd.CTCLoss.apply(x, lab, il, tl)
    assert loss.dim() == 0 and loss.dtype == torch.float32
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 1e-4
    assert np.abs(np_(x.grad) - d["grad"]).max() < GRAD_ATOL_REF
    m = ctc_amd.NoBlankCTC().to(dev)
    x2 = torch.tensor(d["x"]).to(dev).requires_grad_(True)
    loss2 = m(x2, lab.long(), il.cpu(), tl.cpu())  # int64 labels, CPU lengths
    (3.0 * loss2).backward()                       # upstream gradient != 1
    assert abs(float(loss2) - float(d["loss"])) < 1e-4
    assert np.abs(np_(x2.grad) - 3.0 * d["grad"]).max() < 3 * GRAD_ATOL_REF
    with torch.no_grad():
        assert abs(float(m(x2, lab, il, tl)) - float(d["loss"])) < 1e-4


def test_noblank_noncontiguous_and_retain_graph(dev):
    import ctc_amd
    x, lab, Tb, L = synth_noblank(5, 12, 6, 9, 4, var_T=True)
    ref = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    xb = x.permute(1, 0, 2).contiguous().to(dev)          # [B,T,C] storage
    xv = xb.permute(1, 0, 2).requires_grad_(True)         # [T,B,C] view, strides (C, T*C, 1)
    loss, _ = ctc_amd.noblank_ctc_loss(xv, lab.to(dev), Tb.to(dev), L.to(dev))
    loss.backward(retain_graph=True)
    g1 = np_(xv.grad).copy()
    xv.grad = None
    loss.backward()
    assert np.abs(g1 - ref["grad"]).max() < 2e-6 and np.abs(np_(xv.grad) - ref["grad"]).max() < 2e-6


@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (10, 10, 33, 10), (37, 5, 64, 7), (16, 3, 65, 16),
                                   (50, 6, 200, 50), (40, 3, 300, 12), (130, 2, 40, 100), (150, 8, 158, 20)])
@pytest.mark.parametrize("var_T", [False, True])
def test_noblank_vs_oracle_shapes(dev, shape, var_T):
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(sum(shape), T, B, C, S, var_T=var_T, int64=bool(T % 2))
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert_close(r, ref, 2e-6 * max(1.0, 256.0 / B))


def test_noblank_config2_full_size(dev):
    """B=256 T=150 C=158 S<=20: float32 oracle (reference op order) and float64 truth."""
    import ctc_amd
    x, lab, Tb, L = synth_noblank(0, 150, 256, 158, 20)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    ref32 = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float32, threads=8)
    ref64 = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64, threads=8)
    # nll: the float32 oracle keeps the reference's rounding sequence; gradient: float64
    # (the float32 oracle's own exp(alpha+beta+nll) is only good to ~1e-3 relative)
    assert (np.abs(r["nll"] - ref32["nll"]) <= NLL_RTOL * np.abs(ref32["nll"])).all()
    assert_close(r, ref64, 2e-6)
    # properties that hold at any size: every live row of the gradient sums to zero
    # (softmax minus a distribution), dead rows are zero, mean of nll is the loss
    assert np.abs(r["grad"].sum(axis=2)).max() < 1e-7
    assert abs(r["nll"].mean() - r["loss"]) < 1e-3
    # sharding property (config 4): a shard scaled by 1/B_global reproduces its slice
    xs = x[:, 64:128].contiguous()
    rs = run_hip(ctc_amd.noblank_ctc_loss, xs, lab[64:128], Tb[64:128], L[64:128], dev, batch_total=256)
    assert np.abs(rs["grad"] - r["grad"][:, 64:128]).max() < 1e-9
    assert np.abs(rs["nll"] - r["nll"][64:128]).max() == 0.0


def test_noblank_infeasible_and_deterministic(dev):
    import ctc_amd
    x, lab, Tb, L = synth_noblank(9, 10, 4, 6, 8)
    L[:] = torch.tensor([8, 2, 8, 1]); Tb[:] = torch.tensor([10, 10, 5, 10])   # sample 2: L > T_b
    lab = torch.randint(0, 6, (4, 8), dtype=torch.int32)
    r1 = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    r2 = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert r1["nll"][2] >= 1e12 and np.abs(r1["grad"][:, 2]).max() == 0.0
    assert (r1["grad"] == r2["grad"]).all() and (r1["nll"] == r2["nll"]).all() and r1["loss"] == r2["loss"]
    ref = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    ok = [0, 1, 3]
    assert np.abs(r1["nll"][ok] - ref["nll"][ok]).max() < 1e-4
    assert np.abs(r1["grad"][:, ok] - ref["grad"][:, ok]).max() < 1e-6


# ------------------------------------------------------------------ binary
@pytest.mark.parametrize("name", ["kat2_binary", "cfg1_binary", "charades_binary", "edge_binary"])
def test_binary_golden(golden, dev, name):
    import ctc_amd
    d = golden(name)
    r = run_hip(ctc_amd.binary_ctc_loss, d["x"], d["y"], d["in_len"], d["tgt_len"], dev)
    assert_close(r, d, 1e-5)
    for b, tb in enumerate(d["in_len"]):
        assert np.abs(r["grad"][int(tb):, b]).max(initial=0.0) == 0.0


def test_binary_modules_and_function_surface(golden, dev):
    import ctc_amd
    d = golden("cfg1_binary")
    x = torch.tensor(d["x"]).to(dev).requires_grad_(True)
    y = torch.tensor(d["y"]).to(dev)
    il, tl = torch.tensor(d["in_len"]).to(dev), torch.tensor(d["tgt_len"]).to(dev)
    loss = ctc_amd.CTCLoss.apply(x, y, il, tl)              # float [B,S,C] targets -> binary variant
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 1e-4
    assert np.abs(np_(x.grad) - d["grad"]).max() < 1e-5
    x2 = torch.tensor(d["x"]).to(dev).requires_grad_(True)
    loss2 = ctc_amd.NoBlankBinaryCTC()(x2, y, il.cpu(), tl.cpu())
    (0.5 * loss2).backward()
    assert np.abs(np_(x2.grad) - 0.5 * d["grad"]).max() < 1e-5


@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (10, 10, 33, 10), (37, 5, 64, 7), (16, 3, 65, 16),
                                   (50, 6, 200, 50), (40, 3, 300, 12), (60, 2, 40, 100), (150, 8, 158, 20)])
@pytest.mark.parametrize("var_T", [False, True])
def test_binary_vs_oracle_shapes(dev, shape, var_T):
    import ctc_amd
    T, B, C, S = shape
    x, y, Tb, L = synth_binary(sum(shape), T, B, C, S, var_T=var_T, density=0.1)
    if T % 2:
        y = y * torch.rand(y.shape, generator=torch.Generator().manual_seed(1))   # soft targets
    ref = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    assert_close(r, ref, 2e-7 * max(1.0, 256.0 / B))


def test_binary_config3_full_size(dev):
    import ctc_amd
    x, y, Tb, L = synth_binary(0, 150, 256, 158, 20)
    r = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    ref32 = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float32, threads=8)
    ref64 = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float64, threads=8)
    assert (np.abs(r["nll"] - ref32["nll"]) <= NLL_RTOL * np.abs(ref32["nll"])).all()
    assert_close(r, ref64, 1e-7)
    # large |x| (reference parity domain |x| <= 15: the -100 clamp / p==1 rounding are mimicked)
    xb = (x * 4).clamp(-15, 15)
    r = run_hip(ctc_amd.binary_ctc_loss, xb, y, Tb, L, dev)
    ref32 = ctc_c.binary_ctc(np_(xb), np_(y), np_(Tb), np_(L), np.float32, threads=8)
    assert (np.abs(r["nll"] - ref32["nll"]) <= 1e-4 * np.abs(ref32["nll"])).all()
    assert np.abs(r["grad"] - ref32["grad"]).max() < 1e-6


# ------------------------------------------------------------------ blank CTC (config 5 semantics)
def _torch_ctc(lp, tgt, Tb, L):
    """nll from torch fp32; gradient from torch float64: torch's fp32 backward evaluates
    exp(alpha+beta+nll-lp) at magnitudes of ~1e3, which is only good to ~3e-4 relative
    (measured 1.6e-4 abs at T=200, L=1) -- the HIP path normalises per lattice row instead."""
    lpc = lp.double().clone().requires_grad_(True)
    loss = torch.nn.functional.ctc_loss(lpc, tgt, Tb, L, blank=0, reduction="mean", zero_infinity=False)
    loss.backward()
    nll = torch.nn.functional.ctc_loss(lp, tgt, Tb, L, blank=0, reduction="none")
    return {"loss": float(loss), "nll": np_(nll), "grad": np_(lpc.grad)}


@pytest.mark.parametrize("name", ["blank_small", "blank_long"])
def test_blank_golden(golden, dev, name):
    import ctc_amd
    d = golden(name)
    r = run_hip(ctc_amd.blank_ctc_loss, d["lp"], d["tgt"], d["in_len"], d["tgt_len"], dev)
    assert_close(r, d, 1e-5)
    m = ctc_amd.BlankCTC(blank=0)
    lp = torch.tensor(d["lp"]).to(dev).requires_grad_(True)
    loss = m(lp, torch.tensor(d["tgt"]).to(dev), torch.tensor(d["in_len"]), torch.tensor(d["tgt_len"]))
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 1e-4 and np.abs(np_(lp.grad) - d["grad"]).max() < 1e-5


@pytest.mark.parametrize("shape", [(21, 2, 16, 5), (50, 4, 20, 8), (30, 3, 7, 30), (64, 5, 300, 31),
                                   (90, 3, 40, 63), (120, 2, 50, 64), (200, 2, 1000, 100), (300, 2, 30, 140),
                                   (600, 3, 24, 30),      # long sequences; 2 states per lane (4 and 8 are the
                                   (1100, 3, 16, 40)])    # two shapes before)
@pytest.mark.parametrize("var_T", [False, True])
@pytest.mark.parametrize("schedule", ["auto", "persistent", "pool"])
def test_blank_vs_torch_cpu(dev, shape, var_T, schedule, monkeypatch):
    """the third-party arithmetic itself (torch CPU F.ctc_loss) is the comparator here.  `persistent` forces
    the single persistent launch of blank.hip (the library picks it by itself only for config-5-like batches),
    `pool` forces it with the worker pool gathering the emission rows (float4-able rows: C % 4 == 0)"""
    import ctc_amd
    T, B, C, S = shape
    if schedule in ("persistent", "pool"):
        if T < 128 or (schedule == "pool" and C % 4):
            pytest.skip("the persistent launch needs T >= 128 (the pool gather: float4 rows)")
        _schedule(monkeypatch, 1 if schedule == "persistent" else 2)
    else:
        _schedule(monkeypatch, -1)
    lp, tgt, Tb, L = synth_blank(sum(shape), T, B, C, S, var_T=var_T)
    if not var_T:
        tgt[0, 1:4] = tgt[0, 0]                   # repeated labels force blanks in between
        L[0] = min(S, max(int(L[0]), 4))
        L[-1] = 1
    feasible = [b for b in range(B) if int(Tb[b]) >= int(L[b]) + int((tgt[b, 1:int(L[b])] == tgt[b, :int(L[b]) - 1]).sum())]
    ref = _torch_ctc(lp, tgt, Tb, L)
    r = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
    assert len(feasible) >= 1
    fb = np.array(feasible)
    assert (np.abs(r["nll"][fb] - ref["nll"][fb]) <= 1e-5 * np.maximum(1, np.abs(ref["nll"][fb]))).all()
    # (fp32 scans in the log domain: the error of a posterior grows with the length of the sequence)
    gerr = np.abs(r["grad"][:, fb] - ref["grad"][:, fb]).max()
    print("blank %s %s var_T=%s: max grad err %.2e" % (shape, schedule, var_T, gerr))
    assert gerr < min(1e-4, 2e-6 * max(1.0, 64.0 / B) * max(1.0, T / 300.0))   # never looser than north_star's 1e-4
    for b in range(B):
        if b not in feasible:
            assert np.isinf(r["nll"][b]) and np.isinf(ref["nll"][b])
            assert np.abs(r["grad"][:, b]).max() == 0.0     # documented: zero, where torch gives NaN


@pytest.mark.parametrize("schedule", ["1", "0", "2"])
def test_blank_fused_schedule_edge_cases(dev, schedule, monkeypatch):
    """the single persistent launch (and the three launches): ragged lengths, a one-frame sample, an empty
    target, a sample without any alignment and an empty input in one batch, against the float64 oracle"""
    import ctc_amd
    _schedule(monkeypatch, int(schedule))
    T, B, C, S = 160, 7, 36, 20
    lp, tgt, Tb, L = synth_blank(77, T, B, C, S, var_T=True)
    Tb[0], L[0] = T, S
    Tb[1], L[1] = 1, 1
    Tb[2], L[2] = T, 0
    Tb[3], L[3] = 5, 9                                # no alignment: inf, zero gradient
    Tb[4], L[4] = 0, 0
    tgt[5, 0:6] = tgt[5, 0]                           # a run of equal labels: blanks forced between them
    Tb[5], L[5] = T - 3, 8
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64)
    for targets in (tgt, tgt.int()):
        r = run_hip(ctc_amd.blank_ctc_loss, lp, targets, Tb, L, dev)
        fin = np.isfinite(ref["nll"])
        assert (np.isinf(r["nll"]) == ~fin).all() and not fin[3] and fin.sum() == B - 1
        assert np.abs(r["nll"][fin] - ref["nll"][fin]).max() <= 1e-5 * max(1.0, np.abs(ref["nll"][fin]).max())
        assert np.abs(r["grad"][:, fin] - ref["grad"][:, fin]).max() < 2e-6 * 64.0 / B
        assert np.abs(r["grad"][:, 3]).max() == 0.0 and np.abs(r["grad"][1:, 1]).max() == 0.0
    # the same rows again through a strided view (the loaders and workers honour the strides)
    wide = torch.randn(T, B, C + 12)
    wide[:, :, 4:4 + C] = lp
    xv = wide.to(dev)[:, :, 4:4 + C].requires_grad_(True)
    loss, nll = ctc_amd.blank_ctc_loss(xv, tgt.to(dev), Tb.to(dev), L.to(dev))
    loss.backward()
    assert np.abs(np_(xv.grad)[:, fin] - ref["grad"][:, fin]).max() < 2e-6 * 64.0 / B


@pytest.mark.parametrize("T", [129, 160, 257])
def test_blank_persistent_launch_row_pairs(dev, T, monkeypatch):
    """The persistent launch keeps every other lattice row and its workers take rows in pairs (2P, 2P+1): odd and even T,
    odd and even T_b (an odd T_b leaves its last row without a partner), T_b = 1, 2, 3, T-1, T, lengths that put the
    chains' crossing on either parity, two / four / eight states per lane -- both schedules against the float64 oracle
    and against each other."""
    import ctc_amd
    for S, C in ((20, 36), (100, 512), (200, 64)):
        B = 10
        lp, tgt, Tb, L = synth_blank(5 * T + S, T, B, C, S, var_T=True)
        Tb[:8] = torch.tensor([1, 2, 3, T - 1, T, T // 2, T // 2 + 1, 7])
        L[:8] = torch.tensor([1, 1, 2, min(S, (T - 1) // 2), min(S, T // 2), min(S, T // 4), 3, 3])
        ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64, threads=8)
        fin = np.isfinite(ref["nll"])
        assert fin[:8].all()
        got = {}
        for schedule in (1, 0, 2):                            # (2: the worker pool gathers the emission rows)
            _schedule(monkeypatch, schedule)
            r = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
            assert (np.isinf(r["nll"]) == ~fin).all()
            assert np.abs(r["nll"][fin] - ref["nll"][fin]).max() <= 1e-5 * max(1.0, np.abs(ref["nll"][fin]).max())
            assert np.abs(r["grad"][:, fin] - ref["grad"][:, fin]).max() < 2e-6 * 64.0 / B
            for b in range(B):                                 # rows beyond T_b: exactly zero
                assert np.abs(r["grad"][int(Tb[b]):, b]).max(initial=0.0) == 0.0
            got[schedule] = r
        assert np.abs(got[1]["grad"] - got[0]["grad"]).max() < 2e-6 * 64.0 / B
        assert np.abs(got[2]["grad"] - got[1]["grad"]).max() == 0.0    # same chains, same rows: bit for bit


@pytest.mark.parametrize("shape", [(40, 3, 12, 6), (150, 3, 1300, 12)])   # three launches; persistent launch with rows
def test_blank_nonzero_blank_index_and_wide_rows(dev, shape, monkeypatch):  # too wide for the float4 path (C > 1024)
    """blank = C-1 instead of 0 (torch CPU as the comparator), targets drawn from the other classes"""
    import ctc_amd
    T, B, C, S = shape
    _schedule(monkeypatch, 1)
    lp, tgt, Tb, L = synth_blank(11 + T, T, B, C, S, var_T=True)
    blank = C - 1
    tgt = (tgt - 1).clamp(min=0)                       # synth_blank draws from 1..C-1: shift to 0..C-2
    lpc = lp.double().clone().requires_grad_(True)
    loss = torch.nn.functional.ctc_loss(lpc, tgt, Tb, L, blank=blank, reduction="mean", zero_infinity=True)
    loss.backward()
    nll = torch.nn.functional.ctc_loss(lp, tgt, Tb, L, blank=blank, reduction="none")
    x = lp.to(dev).requires_grad_(True)
    got, got_nll = ctc_amd.blank_ctc_loss(x, tgt.to(dev), Tb.to(dev), L.to(dev), blank=blank)
    got.backward()
    fin = np.isfinite(np_(nll))
    assert fin.any()
    assert (np.abs(np_(got_nll)[fin] - np_(nll)[fin]) <= 1e-5 * np.maximum(1, np.abs(np_(nll)[fin]))).all()
    assert np.abs(np_(x.grad)[:, fin] - np_(lpc.grad)[:, fin]).max() < 2e-6 * 64.0 / B
    m = ctc_amd.BlankCTC(blank=blank)
    # (a forward-only call takes the three launches and their log2-domain chains: the same number to fp32 rounding)
    assert abs(float(m(lp.to(dev), tgt.to(dev), Tb, L)) - float(got)) <= 2e-6 * max(1.0, abs(float(got)))


def test_blank_persistent_launch_is_the_default_for_config5_like_batches(dev, monkeypatch):
    """B = #CUs/11 .. #CUs/2, 4 states per lane, float4 rows, T >= 256: the library takes the persistent launch by
    itself; the result must agree with the three launches to rounding and with the float64 oracle"""
    import ctc_amd
    T, B, C, S = 260, 32, 512, 100
    lp, tgt, Tb, L = synth_blank(5, T, B, C, S, var_T=True)
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64, threads=8)
    _schedule(monkeypatch, -1)
    auto = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
    _schedule(monkeypatch, 0)
    three = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
    for r in (auto, three):
        assert (np.abs(r["nll"] - ref["nll"]) <= 1e-5 * np.maximum(1, np.abs(ref["nll"]))).all()
        assert np.abs(r["grad"] - ref["grad"]).max() < 2e-6 * 64.0 / B
    d = np.abs(auto["grad"] - three["grad"]).max()
    assert d < 4e-6
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if 11 * B >= cus and 2 * B <= cus:                # (the rule of run_blank, blank.hip)
        assert d > 0.0          # different schedules (beta is stored without its emission): close, not identical


def test_blank_int32_targets_and_oracle(dev):
    import ctc_amd
    lp, tgt, Tb, L = synth_blank(3, 80, 6, 25, 12, var_T=True)
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.blank_ctc_loss, lp, tgt.int(), Tb, L, dev)
    assert_close(r, ref, 1e-6)


# ------------------------------------------------------------------ best path
@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (150, 16, 158, 20), (40, 3, 300, 12), (90, 2, 40, 70)])
def test_noblank_best_path(dev, shape):
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(sum(shape) + 1, T, B, C, S, var_T=True)
    path, score = ctc_amd.noblank_best_path(x.to(dev), lab.to(dev), Tb.to(dev), L.to(dev))
    torch.cuda.synchronize()
    path, score = np_(path), np_(score)
    rp, rs = ctc_numpy.noblank_best_path(np_(x), np_(lab), np_(Tb), np_(L))
    assert np.abs(score - rs).max() <= 1e-5 * np.abs(rs).max()
    for b in range(B):
        tb, l = int(Tb[b]), int(L[b])
        p = path[b]
        assert (p[tb:] == -1).all() and p[0] == 0 and p[tb - 1] == l - 1
        d = np.diff(p[:tb])
        assert ((d == 0) | (d == 1)).all()                      # monotone, one label per step
    # the returned alignment really attains the optimal score (robust to fp near-ties)
    got = ctc_numpy.path_score(np_(x), np_(lab), path, np_(Tb))
    assert np.abs(got - rs).max() <= 1e-4
    assert (path == rp).mean() > 0.98


# ------------------------------------------------------------------ kernel-path boundaries
@pytest.mark.parametrize("shape", [
    (1, 1, 1, 1),        # smallest possible
    (1, 3, 5, 1),        # T = 1
    (165, 3, 70, 9),     # pipelined kernel beyond 160 rows (<= 168)
    (168, 2, 33, 10),    # pipelined kernel, last supported T
    (169, 2, 33, 10),    # phase-serial kernel, two row passes
    (400, 2, 40, 12),    # phase-serial kernel, three row passes
    (30, 9, 256, 64),    # S = 64 (no idle lane: shr/shl chain form), C = 256
    (30, 2, 257, 65),    # S = 65 -> 2 states per lane, C > 256 -> strided rows
    (12, 2, 20, 256),    # S = 256 -> 4 states per lane (L_b <= T_b limits the live states)
])
def test_noblank_kernel_path_boundaries(dev, shape):
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(17 + sum(shape), T, B, C, S, var_T=T > 4)
    L = torch.minimum(L, Tb)
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert_close(r, ref, 2e-6 * max(1.0, 256.0 / B))


@pytest.mark.parametrize("shape", [
    (168, 3, 256, 31),   # largest shape of the four-rows-per-wave kernel (noblank_r16.hpp): T, C, S at their limits
    (150, 2, 158, 20),   # config-2 shape
    (9, 2, 2, 5),        # one column pair per row
    (40, 2, 34, 17),     # second pass of states (l >= 16) barely used, last column pair partly masked
    (33, 4, 64, 16),     # exactly one pass of states
])
def test_noblank_r16_kernel_cases(dev, shape):
    """Cases aimed at the four-rows-per-wave kernel: repeated labels (occupancy accumulation in
    several passes), a single label, T_b = L_b (one alignment), dead tail rows, non-contiguous logits."""
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(5 + sum(shape), T, B, C, S, var_T=True)
    L = torch.minimum(L, Tb)
    lab[0, :] = 1 % C                                           # every state the same class
    if B > 1:
        L[1] = 1                                                # a single state
    if B > 2:
        Tb[2] = L[2]                                            # exactly one alignment
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert_close(r, ref, 2e-6 * max(1.0, 256.0 / B))
    for b, tb in enumerate(np_(Tb)):
        assert np.abs(r["grad"][int(tb):, b]).max(initial=0.0) == 0.0
    # the same through a strided view (every second sample of a wider buffer: strides stay even)
    wide = torch.zeros(T, 2 * B, C)
    wide[:, ::2] = x
    xd = wide.to(dev)[:, ::2].requires_grad_(True)
    loss, nll = ctc_amd.noblank_ctc_loss(xd, lab.to(dev), Tb.to(dev), L.to(dev))
    assert np.array_equal(np_(nll), r["nll"])


def test_noblank_r16_small_shape_sweep(dev):
    """Seeded sweep over small shapes the four-rows-per-wave kernel accepts (C even): every T parity
    (the beta chain peels one step when T_b is even), T_b from 1, single states, S up to 31."""
    import ctc_amd
    rng = np.random.RandomState(7)
    for case in range(48):
        T = int(rng.randint(1, 41))
        B = int(rng.randint(1, 5))
        C = 2 * int(rng.randint(1, 40))
        S = int(rng.randint(1, 32))
        x, lab, Tb, L = synth_noblank(1000 + case, T, B, C, S, var_T=False)
        Tb = torch.from_numpy(rng.randint(1, T + 1, size=B)).long()
        L = torch.minimum(torch.minimum(L, Tb), torch.tensor(S))
        ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
        r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
        assert_close(r, ref, 2e-6 * 256.0 / B)


def test_binary_pipelined_kernel_shape_sweep(dev):
    """Seeded sweep over shapes the pipelined binary kernel takes (S <= 64, T <= 168, C <= 256, images in LDS): every
    number of rounds and emission tiles, 1..4 column chunks, 1..4 label tiles, ragged T_b (down to L_b), soft
    targets on every other case, scaled logits on every third (rows that leave the cheap range)."""
    import ctc_amd
    rng = np.random.RandomState(11)
    for case in range(40):
        T = int(rng.randint(1, 169))
        B = int(rng.randint(1, 5))
        C = int(rng.randint(1, 257))
        S = int(rng.randint(1, 65))
        if (3 * T + 8) * ((S + 3) // 4 * 4) + 96 + T + (S + T + 4) * (C + 40) > 40000:   # keep the images inside 160 KB
            C = min(C, 64)
        x, y, Tb, L = synth_binary(2000 + case, T, B, C, S, var_T=True, density=0.15)
        L = torch.minimum(L, Tb)
        if case % 2:
            y = y * torch.rand(y.shape, generator=torch.Generator().manual_seed(case))
        if case % 3 == 0:
            x = x * 5.0
        ref = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float64)
        ref32 = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float32)
        r = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
        tol = max(2e-7 * max(1.0, 256.0 / B), 2.0 * np.abs(ref32["grad"] - ref["grad"]).max())
        assert np.isfinite(r["grad"]).all(), (case, T, B, C, S)
        assert np.abs(r["grad"] - ref["grad"]).max() <= tol, (case, T, B, C, S)
        # (nll against the float32 port: with |x| up to 20 the reference's own fp32 arithmetic -- p rounding to 1, the
        # -100 clamp -- is 1 % away from float64, and that arithmetic is what the kernel reproduces)
        assert (np.abs(r["nll"] - ref32["nll"]) <= 1e-4 * np.maximum(1.0, np.abs(ref32["nll"]))).all(), (case, T, B, C, S)
        if case % 3:
            assert (np.abs(r["nll"] - ref["nll"]) <= 3e-5 * np.maximum(1.0, np.abs(ref["nll"]))).all(), (case, T, B, C, S)


def test_blank_persistent_launch_shape_sweep(dev, monkeypatch):
    """Seeded sweep with the persistent launch forced: T from its minimum up, ragged T_b and L_b, 2 / 4 / 8 states per
    lane, float4 and scalar rows -- against the float64 oracle."""
    import ctc_amd
    _schedule(monkeypatch, 1)
    rng = np.random.RandomState(13)
    for case in range(12):
        T = int(rng.randint(128, 400))
        B = int(rng.randint(1, 9))
        S = int([12, 40, 100, 130, 250][rng.randint(0, 5)])
        C = int([37, 64, 260, 1001][rng.randint(0, 4)])
        lp, tgt, Tb, L = synth_blank(3000 + case, T, B, C, S, var_T=True)
        ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64, threads=8)
        fin = np.isfinite(ref["nll"])
        r = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
        assert (np.isinf(r["nll"]) == ~fin).all(), (case, T, B, C, S)
        if fin.any():
            assert np.abs(r["nll"][fin] - ref["nll"][fin]).max() <= 1e-5 * max(1.0, np.abs(ref["nll"][fin]).max()), (case, T, B, C, S)
            assert np.abs(r["grad"][:, fin] - ref["grad"][:, fin]).max() < min(1e-4, 2e-6 * 64.0 / B * max(1.0, T / 300.0)), (case, T, B, C, S)


def test_noblank_extreme_logits_keep_full_range(dev):
    """State contrasts far beyond fp32 range (logits x 200: per-sample nll ~ 1e5) -- the lattice
    cells carry their own exponents, so nothing underflows and nothing is approximated."""
    import ctc_amd
    x, lab, Tb, L = synth_noblank(3, 150, 16, 158, 20, var_T=True)
    x = x * 200.0
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert np.isfinite(r["nll"]).all() and (r["nll"] > 1e3).all()
    assert_close(r, ref, 2e-5 * (256.0 / 16), nll_rtol=1e-6)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (170, 2, 40, 12), (160, 2, 158, 20), (30, 2, 257, 6),
                                   (30, 3, 64, 64), (20, 2, 30, 70),
                                   # the pipelined kernel's own edges: T = 168 / 169, fewer than six rounds, odd T,
                                   # four column chunks with four label tiles, a lone emission tile
                                   (168, 2, 40, 12), (169, 2, 40, 12), (161, 3, 158, 20), (29, 2, 256, 64),
                                   (3, 2, 20, 3), (28, 3, 63, 63), (15, 2, 130, 17), (57, 2, 129, 33)])
def test_binary_kernel_path_boundaries(dev, shape):
    import ctc_amd
    T, B, C, S = shape
    x, y, Tb, L = synth_binary(23 + sum(shape), T, B, C, S, var_T=T > 4, density=0.2)
    L = torch.minimum(L, Tb)
    ref = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    assert_close(r, ref, 2e-7 * max(1.0, 256.0 / B))


def test_binary_rows_with_and_without_tails(dev):
    """The pipelined kernel takes the cheap form of the logs per ROW (every element in (-16, 6)) and the careful
    one otherwise: a batch that mixes both kinds of rows, including |x| > 27 (floored BCE denominator), +-inf-ish
    logits and rows that are entirely in the tail, against the float64 oracle and the float32 port."""
    import ctc_amd
    T, B, C, S = 150, 12, 158, 20
    x, y, Tb, L = synth_binary(5, T, B, C, S, var_T=True, density=0.1)
    g = torch.Generator().manual_seed(9)
    rows = torch.rand(T, B, generator=g) < 0.3                     # 30 % of the rows get tails
    bump = torch.where(torch.rand(T, B, C, generator=g) < 0.05, torch.randn(T, B, C, generator=g) * 12.0,
                       torch.zeros(T, B, C))
    x = x + bump * rows[:, :, None]
    x[7, 0] = x[7, 0] * 10.0                                       # a row that is all tail
    x[8, 1, :5] = torch.tensor([-16.0, 6.0, -15.999, 5.999, 30.0]) # the edges of the cheap range
    ref64 = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float64, threads=8)
    ref32 = ctc_c.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float32, threads=8)
    r = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    assert np.isfinite(r["grad"]).all()
    assert (np.abs(r["nll"] - ref32["nll"]) <= 1e-4 * np.abs(ref32["nll"])).all()
    # (the float32 arithmetic of the reference itself is what limits the agreement with float64 in the tails)
    err32 = np.abs(ref32["grad"] - ref64["grad"]).max()
    assert np.abs(r["grad"] - ref64["grad"]).max() <= max(2e-6, 2.0 * err32)


def test_unsupported_shapes_raise(dev):
    import ctc_amd
    # shapes the kernels do not tile are reported by the C ABI as "unsupported shape" and the
    # Python layer raises -- never a silent fallback
    x, y, Tb, L = synth_binary(1, 3000, 1, 8, 20)            # binary lattice must fit in LDS
    with pytest.raises(ctc_amd.CtcAmdError):
        ctc_amd.binary_ctc_loss(x.to(dev), y.to(dev), Tb, L)
    x, lab, Tb, L = synth_noblank(1, 4, 1, 8, 300)
    with pytest.raises(ctc_amd.CtcAmdError):
        ctc_amd.noblank_ctc_loss(x.to(dev), lab.to(dev), Tb, torch.minimum(L, Tb))
    with pytest.raises(ValueError):
        ctc_amd.noblank_ctc_loss(x.to(dev).double(), lab.to(dev), Tb, L)
    with pytest.raises(ValueError):
        ctc_amd.noblank_ctc_loss(x.to(dev), lab.to(dev), torch.tensor([9]), L)     # T_b > T (CPU lengths: checked)


def test_noblank_more_samples_than_cus(dev):
    """B > #CUs selects the 64-VGPR build (two workgroups per CU); B % 8 != 0 exercises the
    XCD-aware sample mapping's remainder handling."""
    import ctc_amd
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    B = ncu + 37
    x, lab, Tb, L = synth_noblank(77, 60, B, 70, 12, var_T=True)
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64, threads=8)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert_close(r, ref, 2e-6)
    xb, y, Tb, L = synth_binary(78, 40, B, 40, 9, var_T=True, density=0.2)
    ref = ctc_c.binary_ctc(np_(xb), np_(y), np_(Tb), np_(L), np.float64, threads=8)
    r = run_hip(ctc_amd.binary_ctc_loss, xb, y, Tb, L, dev)
    assert_close(r, ref, 2e-7)


@pytest.mark.parametrize("B", [8, 600])
def test_autograd_path_is_graph_capturable(dev, B):
    """the C ABI only enqueues work (no allocation, no host sync): forward + backward can be
    captured into a hipGraph and replayed on new data in the same buffers (B = 8: the one-sample-per-workgroup launch;
    B = 600: the persistent form, replayed twice -- nothing of a launch's state may survive it)"""
    import ctc_amd
    x, lab, Tb, L = synth_noblank(5, 30, B, 20, 6, var_T=True)
    xs = x.to(dev).requires_grad_(True)
    labd, Tbd, Ld = lab.to(dev), Tb.to(dev), L.to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                     # warm-up on the capture stream (workspace, grads)
        for _ in range(2):
            xs.grad = None
            loss = ctc_amd.CTCLoss.apply(xs, labd, Tbd, Ld)
            loss.backward()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    xs.grad = None
    with torch.cuda.graph(g):
        loss = ctc_amd.CTCLoss.apply(xs, labd, Tbd, Ld)
        loss.backward()
    for seed in (6, 7):
        x2, _, _, _ = synth_noblank(seed, 30, B, 20, 6)
        with torch.no_grad():
            xs.copy_(x2.to(dev))
        g.replay()
        torch.cuda.synchronize()
        ref = ctc_numpy.noblank_ctc(np_(x2), np_(lab), np_(Tb), np_(L), np.float64)
        assert abs(float(loss.detach()) - float(ref["loss"])) < 1e-4 * max(1.0, abs(float(ref["loss"])))
        assert np.abs(np_(xs.grad) - ref["grad"]).max() < 2e-6 * 32 * 8 / B


@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (150, 8, 158, 20), (40, 3, 300, 70), (168, 5, 64, 31), (61, 300, 34, 9),
                                   (50, 6, 33, 12)])
def test_noblank_posteriors(dev, shape):
    """gamma out of the loss kernel's own chains (the four-rows-per-wave kernel where it takes the shape -- the first, second,
    fourth and fifth case --, the phase-serial kernel otherwise: C > 256, odd C)"""
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(sum(shape) + 3, T, B, C, S, var_T=True)
    gamma, nll = ctc_amd.noblank_posteriors(x.to(dev), lab.to(dev), Tb.to(dev), L.to(dev))
    torch.cuda.synchronize()
    gamma, nll = np_(gamma), np_(nll)
    ref = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    assert np.abs(gamma - ref["gamma"].transpose(1, 0, 2)).max() < 2e-4      # fp32 scans: ~1e-4 relative
    assert (np.abs(nll - ref["nll"]) <= 1e-5 * np.abs(ref["nll"])).all()
    for b in range(B):
        assert np.abs(gamma[b, :int(Tb[b])].sum(axis=1) - 1.0).max() < 1e-5     # a distribution per live step
        assert np.abs(gamma[b, int(Tb[b]):]).max(initial=0.0) == 0.0
        assert np.abs(gamma[b, :, int(L[b]):]).max(initial=0.0) == 0.0


@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (150, 8, 158, 20), (37, 5, 64, 7), (60, 2, 40, 64), (168, 3, 33, 10)])
def test_binary_posteriors(dev, shape):
    """SURVEY 8(f) rank 1 for the binary lattice: gamma out of the pipelined kernel's gradient phase"""
    import ctc_amd
    T, B, C, S = shape
    x, y, Tb, L = synth_binary(sum(shape) + 5, T, B, C, S, var_T=True, density=0.2)
    gamma, nll = ctc_amd.binary_posteriors(x.to(dev), y.to(dev), Tb.to(dev), L.to(dev))
    torch.cuda.synchronize()
    gamma, nll = np_(gamma), np_(nll)
    ref = ctc_numpy.binary_ctc(np_(x), np_(y), np_(Tb), np_(L), np.float64)
    assert np.abs(gamma - ref["gamma"].transpose(1, 0, 2)).max() < 2e-4      # fp32 scans: ~1e-4 relative
    assert (np.abs(nll - ref["nll"]) <= 1e-5 * np.maximum(1.0, np.abs(ref["nll"]))).all()
    for b in range(B):
        assert np.abs(gamma[b, :int(Tb[b])].sum(axis=1) - 1.0).max() < 1e-5     # a distribution per live step
        assert np.abs(gamma[b, int(Tb[b]):]).max(initial=0.0) == 0.0
        assert np.abs(gamma[b, :, int(L[b]):]).max(initial=0.0) == 0.0
    # the loss + gradient call on the same inputs afterwards is unaffected (same workspace)
    r = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    assert np.abs(r["grad"] - ref["grad"]).max() < 2e-6 * max(1.0, 256.0 / B)
    with pytest.raises(ctc_amd.CtcAmdError):                  # beyond the pipelined kernel: says so
        ctc_amd.binary_posteriors(torch.zeros(200, 2, 8, device=dev), torch.zeros(2, 3, 8, device=dev),
                                  torch.tensor([200, 200]), torch.tensor([3, 3]))


@pytest.mark.parametrize("shape", [(2000, 2, 50, 20), (700, 3, 300, 40), (900, 2, 20, 100)])
def test_noblank_long_sequences_use_workspace_lattice(dev, shape):
    """T x S beyond LDS: the lattice moves to the workspace (ctc_amd_workspace_bytes grows)."""
    import ctc_amd
    from ctc_amd import _lib
    T, B, C, S = shape
    assert _lib.load().ctc_amd_workspace_bytes(0, T, B, C, S) > 256
    x, lab, Tb, L = synth_noblank(sum(shape), T, B, C, S, var_T=True)
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
    assert_close(r, ref, 2e-5 * max(1.0, 256.0 / B), nll_rtol=2e-5)     # T ~ 1e3: nll ~ 5e3, ulp 5e-4
    gamma, _ = ctc_amd.noblank_posteriors(x.to(dev), lab.to(dev), Tb.to(dev), L.to(dev))
    assert np.abs(np_(gamma).sum(axis=2)[0, :int(Tb[0])] - 1.0).max() < 1e-4


# ------------------------------------------------------------------ BASELINE configs at full size
@pytest.mark.parametrize("schedule", ["1", "0"])
def test_blank_config5_full_size(dev, schedule, monkeypatch):
    """BASELINE configs[4]: B=64 T=2000 C=1000 S=100, the persistent launch (1) and the three launches (0),
    against the float64 oracle.  The bar is north_star's 1e-4 on the PER-SAMPLE gradient (the batch mean
    carries 1/(B L_b)): the measured error is printed."""
    import ctc_amd
    _schedule(monkeypatch, int(schedule))
    T, B, C, S = 2000, 64, 1000, 100
    lp, tgt, Tb, L = synth_blank(0, T, B, C, S)
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64, threads=16)
    r = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
    assert np.isfinite(ref["nll"]).all() and np.isfinite(r["nll"]).all()
    nerr = (np.abs(r["nll"] - ref["nll"]) / np.maximum(1.0, np.abs(ref["nll"]))).max()
    gerr = np.abs(r["grad"] - ref["grad"]).max()
    per_sample = np.abs(r["grad"] - ref["grad"]).max(axis=(0, 2)) * B * np.maximum(np_(L), 1)
    # what the reference's own arithmetic (torch fp32 on the CPU) makes of the same batch
    lpc = lp.clone().requires_grad_(True)
    torch.nn.functional.ctc_loss(lpc, tgt, Tb, L, blank=0, reduction="mean").backward()
    terr = np.abs(np_(lpc.grad) - ref["grad"]).max()
    print("config 5, schedule %s: max rel nll err %.2e; gradient of the batch-mean loss: max err %.2e "
          "(torch fp32 CPU against the same float64 truth: %.2e); unnormalised per-sample occupancy: %.2e"
          % (schedule, nerr, gerr, terr, per_sample.max()))
    assert nerr <= 1e-5
    # north_star: loss and input gradient within 1e-4 of the reference's output, which is the gradient of
    # mean_b(nll_b / L_b).  (The fp32 log-domain scans lose precision with T -- alpha reaches -2e4 in log2
    # units, one ulp there is 2e-3 -- exactly as the reference's fp32 kernel does; the unnormalised
    # occupancies are good to ~1e-2 at T = 2000, printed above.)
    assert gerr <= 1e-4 and gerr <= 4 * max(terr, 1e-6)
    # size-independent properties: every live row of exp(lp) - occupancy sums to exp-sum - 1 = 0
    assert np.abs(r["grad"].sum(axis=2)).max() < 1e-6


@pytest.mark.parametrize("variant", ["noblank", "binary"])
def test_config4_single_gpu_shape_full_size(dev, variant):
    """B=2048 T=150 C=158 S<=20 (BASELINE configs[3] on one GPU: more samples than CUs) vs float64."""
    import ctc_amd
    T, B, C, S = 150, 2048, 158, 20
    if variant == "noblank":
        x, tg, Tb, L = synth_noblank(0, T, B, C, S)
        ref = ctc_c.noblank_ctc(np_(x), np_(tg), np_(Tb), np_(L), np.float64, threads=16)
        r = run_hip(ctc_amd.noblank_ctc_loss, x, tg, Tb, L, dev)
        tol = 2e-6 * 256.0 / B
    else:
        x, tg, Tb, L = synth_binary(0, T, B, C, S)
        ref = ctc_c.binary_ctc(np_(x), np_(tg), np_(Tb), np_(L), np.float64, threads=16)
        r = run_hip(ctc_amd.binary_ctc_loss, x, tg, Tb, L, dev)
        tol = 1e-7 * 256.0 / B
    print("B=2048 %s: max grad err %.2e (tolerance %.2e)" % (variant, np.abs(r["grad"] - ref["grad"]).max(), tol))
    assert_close(r, ref, tol)
    assert np.abs(r["grad"].sum(axis=2)).max() < 1e-7 if variant == "noblank" else True


@pytest.mark.parametrize("shape,lam", [((150, 700, 158, 20), None), ((60, 530, 64, 31), None), ((37, 600, 34, 9), None),
                                       ((90, 1030, 192, 12), 0.9), ((168, 515, 20, 5), None), ((150, 2100, 158, 20), None)])
def test_noblank_more_samples_than_cus_persistent(dev, shape, lam):
    """B > 2 #CUs: the persistent form of the four-rows-per-wave kernel (one workgroup per CU walks its samples, the next
    sample's rows waiting in registers).  Ragged T_b / L_b, samples without alignment in between, B no multiple of the
    grid, with and without a gradient, smoothed emission; bit-identical to the same samples run 200 at a time (the
    one-sample-per-workgroup form, B <= #CUs) and against the float64 oracle."""
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(sum(shape), T, B, C, S, var_T=True)
    bad = list(range(5, B, 97))                              # no alignment: L_b > T_b
    for b in bad:
        L[b] = min(S, T)
        Tb[b] = max(1, int(L[b]) - 1)
        lab[b, :int(L[b])] = torch.arange(int(L[b]), dtype=lab.dtype) % C
    kw = {} if lam is None else {"label_smoothing": lam}
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev, **kw)
    ok = np.array([b for b in range(B) if b not in bad])
    assert (r["nll"][bad] >= 1e12).all() and np.abs(r["grad"][:, bad]).max() == 0.0
    ref = ctc_numpy.noblank_ctc(np_(x)[:, ok], np_(lab)[ok], np_(Tb)[ok], np_(L)[ok], np.float64, **kw)
    assert (np.abs(r["nll"][ok] - ref["nll"]) <= NLL_RTOL * np.maximum(1.0, np.abs(ref["nll"]))).all()
    assert np.abs(r["grad"][:, ok] * (B / len(ok)) - ref["grad"]).max() < 2e-6 * max(1.0, 64.0 / B) * 4
    # chunks of <= 200 samples take the one-sample-per-workgroup form: same numbers, bit for bit (gradient scaled by the
    # chunk's own 1/B: compare through the per-sample nll and the unscaled rows)
    for lo in range(0, B, 200):
        hi = min(B, lo + 200)
        rc = run_hip(ctc_amd.noblank_ctc_loss, x[:, lo:hi], lab[lo:hi], Tb[lo:hi], L[lo:hi], dev, **kw)
        assert (rc["nll"] == r["nll"][lo:hi]).all()
        assert np.abs(rc["grad"] * ((hi - lo) / B) - r["grad"][:, lo:hi]).max() <= 1e-9
    # forward only
    with torch.no_grad():
        loss, nll = ctc_amd.noblank_ctc_loss(x.to(dev), lab.to(dev), Tb.to(dev), L.to(dev), **kw)
    torch.cuda.synchronize()
    assert (np_(nll) == r["nll"]).all()
    # twice the same
    r2 = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev, **kw)
    assert (r2["grad"] == r["grad"]).all() and r2["loss"] == r["loss"]


def test_sharded_loss_hip_path_single_rank(dev):
    """ShardedCTCLoss with the real (HIP) local compute, 1-rank degenerate mode: a shard of a
    larger global batch reproduces its slice of the full-batch loss and gradient."""
    import ctc_amd
    from ctc_amd.distributed import ShardedCTCLoss, shard_bounds
    T, B, C, S = 150, 64, 158, 20
    x, lab, Tb, L = synth_noblank(3, T, B, C, S, var_T=True)
    ref = ctc_c.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64, threads=8)
    total, grads = 0.0, []
    for rank in range(2):                                    # both shards, one after the other on this GPU
        lo, hi = shard_bounds(B, rank, 2)
        xs = x[:, lo:hi].contiguous().to(dev).requires_grad_(True)
        res = ShardedCTCLoss(global_batch=B, variant="noblank")(xs, lab[lo:hi].to(dev), Tb[lo:hi].to(dev), L[lo:hi].to(dev))
        res.backward()
        total += float(res.value)                            # (no process group: value == local contribution)
        grads.append(np_(xs.grad))
    assert abs(total - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    assert np.abs(np.concatenate(grads, axis=1) - ref["grad"]).max() < 2e-6 * 256.0 / B
    # variant taken from the targets, binary through the same wrapper
    xb, y, Tbb, Lb = synth_binary(4, 40, 6, 30, 5)
    refb = ctc_c.binary_ctc(np_(xb), np_(y), np_(Tbb), np_(Lb), np.float64)
    xd = xb.to(dev).requires_grad_(True)
    res = ShardedCTCLoss(global_batch=6)(xd, y.to(dev), Tbb.to(dev), Lb.to(dev))
    res.backward()
    assert abs(float(res.value) - float(refb["loss"])) < 1e-5 and np.abs(np_(xd.grad) - refb["grad"]).max() < 1e-6


def test_superseded_workspace_stays_valid_for_captured_graphs(dev):
    """A hipGraph captured at one shape keeps replaying correctly after an eager call with a LARGER shape
    has made the hidden workspace grow (the old one must not be freed under the graph)."""
    import ctc_amd
    T, B, C, S = 300, 4, 40, 30                              # blank-CTC: the lattice lives in the workspace
    lp, tgt, Tb, L = synth_blank(11, T, B, C, S)
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xs = lp.to(dev).requires_grad_(True)
        tg, il, tl = tgt.to(dev), Tb.to(dev), L.to(dev)
        for _ in range(2):                                   # warm-up on the capture stream
            xs.grad = None
            loss, _ = ctc_amd.blank_ctc_loss(xs, tg, il, tl)
            loss.backward()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        xs.grad = None
        with torch.cuda.graph(g, stream=side):
            loss, _ = ctc_amd.blank_ctc_loss(xs, tg, il, tl)
            loss.backward()
        # same stream, bigger shape: the cached workspace is superseded by a larger one
        big = synth_blank(12, 2 * T, 2 * B, C, S)
        xb = big[0].to(dev).requires_grad_(True)
        lb, _ = ctc_amd.blank_ctc_loss(xb, big[1].to(dev), big[2].to(dev), big[3].to(dev))
        lb.backward()
        junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(16)]   # reuse whatever was freed
        xs.grad.zero_()
        g.replay()
        torch.cuda.synchronize()
        del junk
    assert abs(float(loss) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    assert np.abs(np_(xs.grad) - ref["grad"]).max() < 2e-6 * 64.0 / B


def test_workspaces_grow_geometrically_and_can_be_released(dev):
    """A loop whose sequences keep getting longer (length-bucketed batches) must not keep one superseded workspace per
    new maximum: a superseding buffer is >= 1.5x its predecessor, so 50 rising shapes leave O(log) buffers; and
    release_workspaces() gives everything back (eager code, no captured graph alive)."""
    import ctc_amd
    from ctc_amd import functional as F
    ctc_amd.release_workspaces(dev)
    B, C, S = 4, 24, 6
    Ts = list(range(40, 540, 10))                            # blank-CTC: the workspace grows with T
    for T in Ts:
        lp, tgt, Tb, L = synth_blank(T, T, B, C, S)
        x = lp.to(dev).requires_grad_(True)
        loss, _ = ctc_amd.blank_ctc_loss(x, tgt.to(dev), Tb.to(dev), L.to(dev))
        loss.backward()
    torch.cuda.synchronize()
    held = [w for (d, _s, v), ws in F._workspaces.items() if d == dev.index and v == 2 for w in ws]
    sizes = [w.numel() for w in held]
    assert 1 <= len(sizes) <= 9, sizes                       # log_1.5(need(530) / need(40)) + 1
    assert all(b >= a + a // 2 for a, b in zip(sizes, sizes[1:])), sizes
    assert sum(sizes) <= 3.2 * sizes[-1], sizes
    # the last result is still right on the superseding buffer
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64)
    assert abs(float(loss) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    freed = ctc_amd.release_workspaces(dev)
    assert freed >= sum(sizes) and not any(d == dev.index for (d, _s, _v) in F._workspaces)
    loss2, _ = ctc_amd.blank_ctc_loss(x.detach(), tgt.to(dev), Tb.to(dev), L.to(dev))     # allocates afresh
    assert abs(float(loss2) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    assert ctc_amd.workspace_status() == 0


def test_blank_persistent_launch_beside_a_busy_stream(dev, monkeypatch):
    """The persistent blank-CTC launch (workgroups waiting for each other) while ANOTHER stream keeps the GPU
    busy with a long kernel: it may be delayed, it must not hang or return poisoned values."""
    import ctc_amd
    _schedule(monkeypatch, 1)
    T, B, C, S = 512, 40, 400, 60
    lp, tgt, Tb, L = synth_blank(5, T, B, C, S)
    ref = ctc_c.blank_ctc(np_(lp), np_(tgt), np_(Tb), np_(L), np.float64, threads=8)
    a = torch.randn(4096, 4096, device=dev)
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        for _ in range(30):
            a = torch.tanh(a @ a) * 0.01                     # a few ms of matrix work on the other stream
    r = run_hip(ctc_amd.blank_ctc_loss, lp, tgt, Tb, L, dev)
    torch.cuda.synchronize()
    assert np.isfinite(r["nll"]).all() and np.isfinite(r["grad"]).all()
    assert (np.abs(r["nll"] - ref["nll"]) <= 1e-5 * np.maximum(1, np.abs(ref["nll"]))).all()
    assert np.abs(r["grad"] - ref["grad"]).max() < 2e-6 * 64.0 / B * max(1.0, T / 300.0)


# ------------------------------------------------------------------ label-smoothed emission
@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (150, 16, 158, 20), (60, 5, 64, 31), (33, 3, 62, 7)])
@pytest.mark.parametrize("lam", [0.9, 0.5, 1.0])
def test_noblank_label_smoothing(dev, shape, lam):
    """NoBlankCTC.py:100-107 (commented sketch; parity unpinned): HIP against the numpy restatement in float64."""
    import ctc_amd
    T, B, C, S = shape
    x, lab, Tb, L = synth_noblank(sum(shape), T, B, C, S, var_T=True)
    ref = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64, label_smoothing=lam)
    r = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev, label_smoothing=lam)
    assert_close(r, ref, 2e-6 * max(1.0, 256.0 / B))
    assert np.abs(r["grad"].sum(axis=2)).max() < 1e-6
    if lam == 1.0:                                           # lambda = 1 is the module as shipped
        plain = run_hip(ctc_amd.noblank_ctc_loss, x, lab, Tb, L, dev)
        assert np.abs(plain["grad"] - r["grad"]).max() < 1e-9 and np.abs(plain["nll"] - r["nll"]).max() < 1e-4


def test_noblank_label_smoothing_unsupported_shapes_raise(dev):
    import ctc_amd
    x, lab, Tb, L = synth_noblank(1, 20, 2, 11, 4)           # odd C: not the four-rows-per-wave kernel
    with pytest.raises(ctc_amd.CtcAmdError):
        ctc_amd.noblank_ctc_loss(x.to(dev), lab.to(dev), Tb.to(dev), L.to(dev), label_smoothing=0.9)


# ------------------------------------------------------------------ target construction
@pytest.mark.parametrize("shape", [(1, 1, 1), (4, 6, 5), (9, 20, 33), (16, 150, 158), (3, 70, 300), (5, 200, 38),
                                   (6, 10, 38), (4, 10, 31), (4, 10, 32), (3, 12, 64)])
@pytest.mark.parametrize("exact", [False, True])
def test_dedup_multihot_targets_bit_exact(dev, shape, exact):
    """integer work: bit-exact against the numpy restatement of charades_ctc_next_pred.py:646-651,663-678 --
    the reference's int32 row codes by default (C <= 64), whole rows with exact_rows"""
    import ctc_amd
    B, S, C = shape
    if C > 64 and not exact:
        with pytest.raises(OverflowError):                    # the reference's 2**o raises at o = 64
            ctc_amd.dedup_multihot_targets(torch.zeros(shape, dtype=torch.int32, device=dev))
        return
    rng = np.random.default_rng(B * 1000 + S)
    base = (rng.random((B, max(1, S // 3), C)) < 0.1).astype(np.int32)
    pick = rng.integers(0, base.shape[1], (B, S))             # many repeats, adjacent and not
    rows = np.take_along_axis(base, pick[:, :, None].repeat(C, 2), axis=1)
    rows[0, S // 2] = 0                                       # an empty row
    if B > 1:
        rows[1] = 0                                           # a clip without any label
    if B > 2 and C > 33 and S > 4:
        rows[2, :5] = 0                                       # rows that differ only in classes >= 32, rows made of them
        rows[2, 0, [1, 33]] = 1; rows[2, 1, [1, C - 1]] = 1; rows[2, 2, C - 2] = 1; rows[2, 3, 2] = 1; rows[2, 4, 1] = 1
    ref, ref_len = ctc_numpy.dedup_multihot_targets(rows, exact_rows=exact)
    out, length = ctc_amd.dedup_multihot_targets(torch.tensor(rows).to(dev), exact_rows=exact)
    torch.cuda.synchronize()
    assert (np_(length) == ref_len).all() and (np_(out) == ref).all()
    # int64 input, same answer; the block feeds the binary loss after the reference's .float()
    out64, len64 = ctc_amd.dedup_multihot_targets(torch.tensor(rows).long().to(dev), exact_rows=exact)
    assert (np_(out64) == ref).all() and (np_(len64) == ref_len).all()
    assert out.dtype == torch.int32 and length.dtype == torch.int64


def test_dedup_multihot_targets_golden(dev, golden):
    """against the fixture captured from the reference's own tensor operations (make_golden.py F6): the int32
    code wraps at class 31 and drops classes >= 32 -- at the reference's default 38 classes rows {1,33}, {1,35},
    {36}, {2}, {1} give TWO targets"""
    import ctc_amd
    f = golden("dedup_targets")
    for C in (5, 30, 31, 32, 33, 38, 64):
        out, length = ctc_amd.dedup_multihot_targets(torch.tensor(f["rows_%d" % C]).to(dev))
        assert (np_(length) == f["len_%d" % C]).all() and (np_(out) == f["out_%d" % C]).all(), C
    out, length = ctc_amd.dedup_multihot_targets(torch.tensor(f["rows_38"]).to(dev))
    assert int(length[1]) == 2
    _, exact = ctc_amd.dedup_multihot_targets(torch.tensor(f["rows_38"]).to(dev), exact_rows=True)
    assert int(exact[1]) == 5


# ------------------------------------------------------------------ the returned loss is a tensor of its own
def test_loss_supports_inplace_arithmetic(dev):
    """`loss /= accum_steps`, `loss *= w`, `loss += aux` on the returned loss (gradient-accumulation loops around
    train.py:427-444): autograd refuses in-place arithmetic on a VIEW handed out by a custom Function, so the
    loss and nll must not be views of one buffer."""
    import ctc_amd
    x, lab, Tb, L = synth_noblank(3, 30, 6, 12, 5, var_T=True)
    ref = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    for fn in (lambda a, b, c, d: ctc_amd.CTCLoss.apply(a, b, c, d), lambda a, b, c, d: ctc_amd.NoBlankCTC()(a, b, c, d),
               lambda a, b, c, d: ctc_amd.noblank_ctc_loss(a, b, c, d)[0]):
        xd = x.to(dev).requires_grad_(True)
        loss = fn(xd, lab.to(dev), Tb.to(dev), L.to(dev))
        assert loss._base is None
        loss /= 2
        loss *= 3.0
        loss += 1.0
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss) - (float(ref["loss"]) * 1.5 + 1.0)) < 1e-4
        assert np.abs(np_(xd.grad) - 1.5 * ref["grad"]).max() < 1e-6


@pytest.mark.parametrize("B", [1, 3, 256])
def test_batch_sum_resolution_on_tiny_losses(dev, B):
    """The in-launch batch sum is fixed point with 27 - ceil(log2 B) fractional bits (include/ctc_amd.h): tiny
    per-sample nll must still come through to within that resolution, not be flushed at 2^-17."""
    import ctc_amd
    T, C, S = 6, 8, 2
    x = torch.full((T, B, C), -6.0)
    x[:3, :, 1] = 6.0                                         # a forced alignment with margin 12: nll ~ 1e-4
    x[3:, :, 5] = 6.0
    lab = torch.tensor([[1, 5]] * B, dtype=torch.int32)
    Tb, L = torch.full((B,), T), torch.full((B,), S)
    ref = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    loss, nll = ctc_amd.noblank_ctc_loss(x.to(dev), lab.to(dev), Tb.to(dev), L.to(dev))
    torch.cuda.synchronize()
    frac = 27 - int(np.ceil(np.log2(B))) if B > 1 else 27
    assert 1e-5 < float(ref["loss"]) < 1e-3
    assert abs(float(loss) - float(np_(nll).astype(np.float64).mean())) <= 2.0 ** -(frac + 1) + 1e-11
    # (the nll itself is fp32 arithmetic on T emissions of ~1 ulp each: absolute error ~1e-6 whatever its size)
    assert abs(float(loss) - float(ref["loss"])) < 5e-6


def test_binary_streamed_kernel_target_forms_and_boundaries(dev):
    """binary_flow.hpp: multi-hot targets and soft targets that are exact in bf16 (0.5, 0.25, 1.5) take the bf16 emission
    tiles, other soft targets the fp32 tiles -- same shapes, all against the float64 oracle; T = 160 is the streamed
    kernel's last length, T = 161 / C = 200 (four column chunks) go to the barrier-phased kernel."""
    import ctc_amd
    for case, (T, B, C, S) in enumerate([(150, 6, 158, 20), (160, 3, 158, 20), (161, 3, 158, 20), (96, 4, 200, 12), (33, 5, 17, 40)]):
        x, y, Tb, L = synth_binary(4000 + case, T, B, C, S, var_T=True, density=0.1)
        g = torch.Generator().manual_seed(case)
        forms = {"multi-hot": y,
                 "bf16-exact soft": y * torch.tensor([0.5, 0.25, 1.0, 1.5])[torch.randint(0, 4, y.shape, generator=g)],
                 "soft": y * torch.rand(y.shape, generator=g)}
        for name, yy in forms.items():
            ref = ctc_c.binary_ctc(np_(x), np_(yy), np_(Tb), np_(L), np.float64)
            r = run_hip(ctc_amd.binary_ctc_loss, x, yy, Tb, L, dev)
            assert np.isfinite(r["grad"]).all(), (name, T, B, C, S)
            assert (np.abs(r["nll"] - ref["nll"]) <= 3e-6 * np.maximum(1.0, np.abs(ref["nll"]))).all(), (name, T, B, C, S)
            assert np.abs(r["grad"] - ref["grad"]).max() <= 2e-7 * max(1.0, 256.0 / B), (name, T, B, C, S)


def test_binary_streamed_kernel_is_deterministic_and_a_shard_is_a_slice(dev):
    """size-independent properties of the streamed binary kernel at config 3's size: two runs agree bit for bit (single-writer
    hand-offs, no atomics on the data path), and a shard of the batch scaled by the global batch size is the slice of the
    full batch's gradient."""
    import ctc_amd
    x, y, Tb, L = synth_binary(21, 150, 256, 158, 20, var_T=True)
    a = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    b = run_hip(ctc_amd.binary_ctc_loss, x, y, Tb, L, dev)
    assert np.array_equal(a["nll"], b["nll"]) and np.array_equal(a["grad"], b["grad"]) and a["loss"] == b["loss"]
    lo, hi = 64, 160
    s = run_hip(ctc_amd.binary_ctc_loss, x[:, lo:hi], y[lo:hi], Tb[lo:hi], L[lo:hi], dev, batch_total=256)
    assert np.array_equal(s["nll"], a["nll"][lo:hi]) and np.array_equal(s["grad"], a["grad"][:, lo:hi])


@pytest.mark.parametrize("shape", [(20, 4, 10, 5), (150, 16, 158, 20), (40, 3, 300, 12), (90, 2, 40, 70)])
def test_binary_best_path(dev, shape):
    """SURVEY 8f-1 on the binary lattice: the Viterbi alignment against the float64 restatement -- optimal score, a
    monotone path from label row 0 to L_b - 1, the returned path attaining the optimum, soft targets on odd T."""
    import ctc_amd
    T, B, C, S = shape
    x, y, Tb, L = synth_binary(sum(shape) + 2, T, B, C, S, var_T=True, density=0.1)
    if S % 2:
        y = y * torch.rand(y.shape, generator=torch.Generator().manual_seed(2))
    path, score = ctc_amd.binary_best_path(x.to(dev), y.to(dev), Tb.to(dev), L.to(dev))
    torch.cuda.synchronize()
    path, score = np_(path), np_(score)
    rp, rs = ctc_numpy.binary_best_path(np_(x), np_(y), np_(Tb), np_(L))
    assert np.abs(score - rs).max() <= 2e-5 * np.abs(rs).max()
    e = ctc_numpy.binary_emissions(np_(x), np_(y))
    for b in range(B):
        tb, l = int(Tb[b]), int(L[b])
        p = path[b]
        assert (p[tb:] == -1).all() and p[0] == 0 and p[tb - 1] == l - 1
        d = np.diff(p[:tb])
        assert ((d == 0) | (d == 1)).all()
        got = e[np.arange(tb), b, p[:tb]].sum()                 # the returned alignment attains the optimum
        assert abs(got - rs[b]) <= 1e-4 * max(1.0, abs(rs[b]))
    assert (path == rp).mean() > 0.97


def test_binary_forward_only_launch_gives_the_same_loss(dev):
    """no gradient requested (validation, `torch.no_grad()`): the streamed kernel leaves after the logs / the tiles / the alpha
    scan -- same per-sample nll and batch mean, bit for bit, as the launch that also produces the gradient"""
    import ctc_amd
    for (T, B, C, S) in [(150, 256, 158, 20), (37, 5, 64, 7), (160, 3, 100, 33)]:
        x, y, Tb, L = synth_binary(1, T, B, C, S, var_T=True)
        xd, yd, Tbd, Ld = x.to(dev), y.to(dev), Tb.to(dev), L.to(dev)
        l0, n0 = ctc_amd.binary_ctc_loss(xd, yd, Tbd, Ld)
        xg = xd.clone().requires_grad_(True)
        l1, n1 = ctc_amd.binary_ctc_loss(xg, yd, Tbd, Ld)
        l1.backward()
        torch.cuda.synchronize()
        assert torch.equal(l0, l1.detach()) and torch.equal(n0, n1), (T, B, C, S)
    assert ctc_amd.workspace_status() == 0
