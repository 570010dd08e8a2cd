"""ctypes binding of libctc_amd.so (the C ABI declared in include/ctc_amd.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CTC_AMD_LIB") or os.path.join(_HERE, "lib", "libctc_amd.so")   # override: kernel experiments

NOBLANK, BINARY, BLANK = 0, 1, 2
ABI_VERSION = 2                     # CTC_AMD_ABI_VERSION of include/ctc_amd.h this binding was written for
ERR_UNSUPPORTED_SHAPE = -2
ERR_CODE_OVERFLOW = -3

_vp, _i64, _int, _f32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes): exactly the prototypes of include/ctc_amd.h
PROTOTYPES = {
    "ctc_amd_abi_version": (_int, []),
    "ctc_amd_error_string": (ctypes.c_char_p, [_int]),
    "ctc_amd_workspace_bytes": (_sz, [_int, _int, _int, _int, _int]),
    "ctc_amd_noblank_loss_grad": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _int, _int,
                                         _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctc_amd_noblank_smoothed_loss_grad": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _int, _int, _f32,
                                                  _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctc_amd_binary_loss_grad": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _int, _int, _int, _int,
                                        _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctc_amd_blank_loss_grad": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _int, _int, _int,
                                       _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctc_amd_scale_grad": (_int, [_vp, _vp, _sz, _vp]),
    "ctc_amd_collective_gate": (_int, [_vp, _int, _int, _vp]),
    "ctc_amd_binary_posteriors": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "ctc_amd_lstm_cell_step": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp,
                                      _vp, _i64, _int, _f32, _vp]),
    "ctc_amd_lstm_series": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int,
                                   _vp, _i64, _i64, _int, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctc_amd_lstm_series_backward": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "ctc_amd_head_forward": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _int, _int, _int, _int,
                                    _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ctc_amd_dedup_multihot_targets": (_int, [_vp, _int, _int, _int, _int, _vp, _vp, _vp]),
    "ctc_amd_blank_set_schedule": (_int, [_int]),
    "ctc_amd_workspace_status": (_int, [_vp, _int, _vp, ctypes.POINTER(ctypes.c_uint)]),
    "ctc_amd_noblank_best_path": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _int, _int,
                                         _vp, _vp, _vp, _vp]),
    "ctc_amd_binary_best_path": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "ctc_amd_noblank_posteriors": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _int, _int,
                                          _vp, _vp, _vp, _vp]),
}

_lib = None


class CtcAmdError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises CtcAmdError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise CtcAmdError(
            "ctc_amd: %s not found -- build it with `python -m ctc_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % SO_PATH)
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.ctc_amd_abi_version() != ABI_VERSION:
        raise CtcAmdError("ctc_amd: %s has ABI version %d, this binding needs %d -- rebuild with "
                          "`python -m ctc_amd.build --force`" % (SO_PATH, lib.ctc_amd_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ctc_amd_error_string(rc)
        raise CtcAmdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
