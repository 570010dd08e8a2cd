"""C-ABI library: loads, exports every symbol include/ctc_amd.h declares; host-side
validation of the Python mirror (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from ctc_amd import build
    return build.build()


def _declared():
    src = open(os.path.join(ROOT, "include", "ctc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctc_amd_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(built)
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), "libctc_amd.so does not export %s" % n


def test_binding_covers_header(built):
    from ctc_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared()
    lib = _lib.load()
    assert lib.ctc_amd_abi_version() == _lib.ABI_VERSION == 2
    assert lib.ctc_amd_workspace_bytes(0, 150, 256, 158, 20) >= 256
    assert lib.ctc_amd_workspace_bytes(2, 2000, 64, 1000, 100) >= 2 * 64 * 2000 * 201 * 4
    assert b"success" in lib.ctc_amd_error_string(0)


def test_argument_errors_without_gpu(built):
    from ctc_amd import _lib
    lib = _lib.load()
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.ctc_amd_noblank_loss_grad(None, 0, 0, None, 0, None, None, 1, 1, 1, 1, 1.0, 1.0,
                                         None, None, None, None, None) == -1
    assert lib.ctc_amd_scale_grad(None, None, 4, None) == -1


def test_no_cpu_fallback():
    import ctc_amd
    x = torch.randn(4, 2, 5, requires_grad=True)
    lab = torch.zeros(2, 3, dtype=torch.long)
    with pytest.raises(ctc_amd.CtcAmdError):
        ctc_amd.CTCLoss.apply(x, lab, torch.tensor([4, 4]), torch.tensor([3, 2]))
    with pytest.raises(ctc_amd.CtcAmdError):
        ctc_amd.NoBlankBinaryCTC()(x, torch.zeros(2, 3, 5), torch.tensor([4, 4]), torch.tensor([3, 2]))


def test_product_code_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ctc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), "%s mentions the oracle" % f


def test_variant_selection_and_shape_errors():
    from ctc_amd import functional as F
    assert F._variant_of(torch.zeros(2, 3, dtype=torch.int32)) == 0
    assert F._variant_of(torch.zeros(2, 3, dtype=torch.int64)) == 0
    assert F._variant_of(torch.zeros(2, 3, 5)) == 1
    with pytest.raises(ValueError):
        F._variant_of(torch.zeros(2, 3))
    with pytest.raises(ValueError):
        F._lengths(torch.tensor([4, 9]), 2, "input_lengths", torch.device("cpu"), 8)
    with pytest.raises(ValueError):
        F._lengths(torch.tensor([0, 3]), 2, "target_lengths", torch.device("cpu"), 8)
    with pytest.raises(ValueError):
        F._lengths(torch.tensor([1, 2, 3]), 2, "target_lengths", torch.device("cpu"), 8)
    assert F._lengths([1, 8], 2, "input_lengths", torch.device("cpu"), 8).dtype == torch.int64
