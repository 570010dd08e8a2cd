"""ctypes front-end of the C oracle (oracle/ctc_oracle.c) -- TEST INFRASTRUCTURE.

Same results dict as oracle/ctc_numpy.py; used for full-size parity checks and as
bench.py's cpu_baseline ("port").  Never imported by ctc_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libctc_oracle.so")
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ctc_oracle.c", "ctc_oracle_impl.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _real(dtype):
    dtype = np.dtype(dtype)
    return ("_f32", ctypes.c_float) if dtype == np.float32 else ("_f64", ctypes.c_double)


def _common(x, in_len, tgt_len, dtype):
    x = np.ascontiguousarray(x, dtype)
    il = np.ascontiguousarray(in_len, np.int64)
    tl = np.ascontiguousarray(tgt_len, np.int64)
    return x, il, tl


def noblank_ctc(x, lab, in_len, tgt_len, dtype=np.float32, scale=None, want_grad=True, threads=1):
    x, il, tl = _common(x, in_len, tgt_len, dtype)
    T, B, C = x.shape
    lab = np.ascontiguousarray(lab, np.int64)
    S = lab.shape[1]
    suf, creal = _real(dtype)
    nll = np.empty(B, dtype)
    grad = np.empty_like(x) if want_grad else None
    fn = getattr(lib(), "oracle_noblank" + suf)
    fn.restype = None
    fn(_p(x), _p(lab), _p(il), _p(tl), T, B, C, S, creal(1.0 / B if scale is None else scale),
       _p(nll), _p(grad), int(threads))
    out = {"nll": nll, "loss": nll.mean(dtype=dtype)}
    if want_grad:
        out["grad"] = grad
    return out


def binary_ctc(x, y, in_len, tgt_len, dtype=np.float32, scale=None, want_grad=True, threads=1):
    x, il, tl = _common(x, in_len, tgt_len, dtype)
    T, B, C = x.shape
    y = np.ascontiguousarray(y, dtype)
    S = y.shape[1]
    suf, creal = _real(dtype)
    nll = np.empty(B, dtype)
    grad = np.empty_like(x) if want_grad else None
    fn = getattr(lib(), "oracle_binary" + suf)
    fn.restype = None
    fn(_p(x), _p(y), _p(il), _p(tl), T, B, C, S, creal(1.0 / B if scale is None else scale),
       _p(nll), _p(grad), int(threads))
    out = {"nll": nll, "loss": nll.mean(dtype=dtype)}
    if want_grad:
        out["grad"] = grad
    return out


def blank_ctc(lp, tgt, in_len, tgt_len, dtype=np.float32, blank=0, want_grad=True, threads=1,
              batch_total=None):
    lp, il, tl = _common(lp, in_len, tgt_len, dtype)
    T, B, C = lp.shape
    tgt = np.ascontiguousarray(tgt, np.int64)
    S = tgt.shape[1]
    suf, _ = _real(dtype)
    nll = np.empty(B, dtype)
    grad = np.empty_like(lp) if want_grad else None
    lens = np.maximum(tl, 1).astype(dtype)
    gs = np.ascontiguousarray(1.0 / ((batch_total or B) * lens), dtype)
    fn = getattr(lib(), "oracle_blank" + suf)
    fn.restype = None
    fn(_p(lp), _p(tgt), _p(il), _p(tl), T, B, C, S, int(blank), _p(gs), _p(nll), _p(grad), int(threads))
    out = {"nll": nll, "loss": (nll / lens).mean(dtype=dtype)}
    if want_grad:
        out["grad"] = grad
    return out
