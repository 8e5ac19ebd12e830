"""The C-ABI library loads and exports every symbol include/icd_search.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from rag_project_icd10_amd import _native


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "icd_search.h"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_native.EXPORTED_SYMBOLS)
    # round 6: no process-wide A/B switch in the product ABI - the former icd_debug_set_* globals are options of ONE index
    # (icd_index_create flags, icd_index_set_option); the library exports none of them, declared or not
    assert not [n for n in names if n.startswith("icd_debug")]
    import subprocess
    dyn = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in dyn.splitlines() if " T " in ln and ln.split()[-1].startswith("icd_")})
    assert exported == names, sorted(set(exported) ^ set(names))
    assert lib.icd_abi_version() == _native.ABI_VERSION == 6


def test_error_reporting_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this case checks the no-GPU error path")
    lib = _native.load_library()
    assert lib.icd_device_count() < 0
    assert b"hipGetDeviceCount" in lib.icd_last_error()
    with pytest.raises(_native.IcdError) as e:
        _native.IcdIndex(np.zeros((4, 768), np.float32))
    assert e.value.code < 0 and "libicdsearch error" in str(e.value)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError) as e:
        _native.load_library(str(tmp_path / "nope.so"))
    assert "no CPU fallback" in str(e.value)


def test_argument_validation_needs_no_gpu():
    lib = _native.load_library()
    out = ctypes.c_void_p()
    buf = np.zeros((2, 64), np.float32)
    assert lib.icd_index_create(None, 2, 64, None, 0, 0, 1, 1, 0, ctypes.byref(out)) == -1   # ICD_ERR_INVALID
    assert lib.icd_index_create(buf.ctypes.data, 2, 50, None, 0, 0, 1, 1, 0, ctypes.byref(out)) == -4  # dim % 32
    assert b"multiple of 32" in lib.icd_last_error()
    assert lib.icd_index_create(buf.ctypes.data, 2, 64, None, 0, 0, 1, 500, 0, ctypes.byref(out)) == -1  # max_k
    assert lib.icd_index_destroy(None) == -5 and lib.icd_index_stats(None, None) == -5
    assert lib.icd_index_set_option(None, 1, 0) == -5
    assert lib.icd_index_create(buf.ctypes.data, 2, 64, None, 0, 0, 1, 1, 1 << 20, ctypes.byref(out)) == -1 and b"unknown bits" in lib.icd_last_error()
    assert lib.icd_merge_topk(0, None, None, None, 1, 1, 1, None, None, None, None, None) == -1


def test_encoder_argument_validation_needs_no_gpu():
    """icd_encoder_*: the checks that need no device - NULL descriptor, shapes the small-input forward is not instantiated for,
    NULL weight pointers, invalid handles - answer with a status and a message before any HIP call"""
    lib = _native.load_library()
    h = ctypes.c_void_p()
    assert lib.icd_encoder_create(0, None, ctypes.byref(h)) == -1 and not h.value          # ICD_ERR_INVALID
    d = _native._EncoderDesc()
    d.layers, d.hidden, d.heads, d.inter, d.vocab, d.max_pos, d.pos_offset, d.ln_eps = 12, 512, 8, 2048, 1000, 512, 0, 1e-12
    assert lib.icd_encoder_create(0, ctypes.byref(d), ctypes.byref(h)) == -4               # ICD_ERR_UNSUPPORTED
    assert b"768 and 1024" in lib.icd_last_error()
    d.hidden, d.heads, d.inter = 768, 12, 1000
    assert lib.icd_encoder_create(0, ctypes.byref(d), ctypes.byref(h)) == -4 and b"inter=1000" in lib.icd_last_error()
    d.inter = 3072
    assert lib.icd_encoder_create(0, ctypes.byref(d), ctypes.byref(h)) == -1 and b"embedding pointer is NULL" in lib.icd_last_error()
    d.heads = 16
    assert lib.icd_encoder_create(0, ctypes.byref(d), ctypes.byref(h)) == -4 and b"64-wide heads" in lib.icd_last_error()
    assert lib.icd_encoder_destroy(None) == -5
    ids = np.array([101, 102], np.int32)
    lens = np.array([2], np.int32)
    out = np.zeros(768, np.float32)
    assert lib.icd_encoder_encode(None, ids.ctypes.data, lens.ctypes.data, 1, 0, 1, out.ctypes.data, 0, None, None) == -5
    assert lib.icd_encoder_encode_many(None, ids.ctypes.data, lens.ctypes.data, 1, 0, 1, out.ctypes.data, 0, None) == -5
    assert not _native.SmallEncoder.supported(object())
