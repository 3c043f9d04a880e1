"""The C-ABI library builds for gfx950, loads, and exports every symbol include/pronto_batch.h declares.
No compute calls here (no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

from pronto_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pronto_batch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libpronto_batch.so does not export %s" % n
    assert set(names) == set(_lib.exported_names()), "python binding out of sync with the header"


def test_code_object_is_gfx950_only():
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_version_and_error_paths_without_device(lib):
    assert lib.pb_version().decode() == "pronto_batch 0.3 gfx950"
    h = C.c_void_p()
    assert lib.pb_create(C.byref(h), 17, 64, 0, 0) == _lib.PB_ERR_ARG          # bad n_states
    assert b"n_states" in lib.pb_last_error(None)
    assert lib.pb_create(C.byref(h), 15, 0, 0, 0) == _lib.PB_ERR_ARG           # bad batch
    assert lib.pb_create(None, 15, 64, 0, 0) == _lib.PB_ERR_ARG
    import torch
    if not torch.cuda.is_available():
        # the product has NO CPU fallback: creation must fail loudly when no GPU is visible
        rc = lib.pb_create(C.byref(h), 15, 64, 0, 0)
        assert rc == _lib.PB_ERR_NO_DEVICE and h.value is None
        assert b"no CPU fallback" in lib.pb_last_error(None)
    assert lib.pb_sync(None) == _lib.PB_ERR_ARG
    assert lib.pb_destroy(None) == _lib.PB_OK


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_binding_refuses_misshaped_arrays_before_calling_the_abi():
    """The C ABI takes no lengths (rows x batch are implied), so the Python binding checks shapes: a wrong batch or row count
    raises ValueError instead of becoming an out-of-bounds host read.  No GPU needed: the check precedes every ABI call."""
    import numpy as np
    import pytest
    from pronto_amd import batch as pa

    class Shapes(pa.BatchEstimator):
        def __init__(self):
            self.B, self.n = 8, 15

        def close(self):
            pass
        __del__ = close

    f = Shapes()
    bad = [(f.predict, (np.zeros((7, 7)), [0, 0, 0, 0])),
           (f.predict, (np.zeros(6), [0, 0, 0, 0])),
           (f.update_indexed, ([3, 4, 5], np.zeros((3, 9)), [1, 1, 1])),
           (f.update_indexed, ([3, 4, 5], np.zeros((3, 8)), np.zeros((3, 7)))),
           (f.update_indexed, ([3, 4, 5], np.zeros((3, 8)), [1, 1, 1], np.ones(7, dtype=np.uint8))),
           (f.step_legodo, (np.zeros((7, 8)), np.zeros((5, 8)), None, [0] * 4)),
           (f.reset, (np.zeros((15, 8)), np.zeros((4, 8)), np.zeros((15, 15, 7)))),
           (f.reset, (np.zeros((21, 8)), np.zeros((4, 8)), np.zeros((15, 15, 8))))]
    for fn, args in bad:
        with pytest.raises(ValueError):
            fn(*args)
