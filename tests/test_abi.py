"""CPU: the C-ABI library builds, loads, and exports every symbol include/rechorus_hip.h
declares; argument validation works without a GPU (no kernels are launched here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from rechorus_amd import _lib

HEADER = os.path.join(ROOT, "include", "rechorus_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from rechorus_amd.csrc.build import build
        build(verbose=False)
    return _lib.load()


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for must in ("rc_gather_dot_fwd", "rc_bpr_loss_fwd_bwd", "rc_bprmf_fwd_bwd", "rc_sort_ids",
                 "rc_segmented_update", "rc_dense_update", "rc_bprmf_train_step"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in rechorus_hip.h but not exported"


def test_python_binding_covers_every_declared_symbol():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_error_reporting(lib):
    assert lib.rc_version() == 1
    rc = lib.rc_gather_rows(None, 64, None, 4, None, None)
    assert rc == -1  # RC_ERR_INVALID_ARG
    assert b"null pointer" in lib.rc_last_error_string()
    with pytest.raises(_lib.RechorusHipError):
        _lib.call("rc_bpr_loss_fwd_bwd", C.c_void_p(8), 4, 1, 0.25, C.c_void_p(8), None, None)
    assert b"C >= 2" in lib.rc_last_error_string()


def test_sasrec_shape_envelope(lib):
    """rc_sasrec_supported is host logic: d in {32, 64}, 1..4 blocks, heads | d, history_max <= 64 on every route; 65..128 with ONE
    block and 1 / 2 / 4 heads (the batch encoder's one-row path).  engine.sasrec_supported adds: no training-mode dropout there."""
    from rechorus_amd import engine
    ok = lambda *a: bool(lib.rc_sasrec_supported(*a))   # (d, n_layers, n_heads, L)
    assert ok(64, 1, 4, 50) and ok(32, 4, 2, 64) and ok(64, 2, 8, 20) and ok(64, 1, 1, 1)
    assert not ok(128, 1, 4, 50) and not ok(64, 5, 4, 50) and not ok(64, 1, 3, 50) and not ok(64, 1, 4, 0)
    assert ok(64, 1, 4, 65) and ok(64, 1, 1, 128) and ok(32, 1, 2, 100) and ok(32, 1, 4, 128)
    assert not ok(64, 2, 4, 65) and not ok(64, 1, 8, 100) and not ok(64, 1, 4, 129)
    assert engine.sasrec_supported(64, 1, 4, 100) and not engine.sasrec_supported(64, 1, 4, 100, dropout=0.1)
    assert engine.sasrec_supported(64, 2, 4, 64, dropout=0.5)


def test_opt_hyper_struct_layout():
    # struct rc_opt_hyper: 2 ints, 5 doubles, 1 int64 -> 56 bytes, natural alignment
    assert C.sizeof(_lib.OptHyper) == 56
    assert _lib.OptHyper.lr.offset == 8 and _lib.OptHyper.step.offset == 48


def test_engine_refuses_cpu_tensors():
    import torch
    from rechorus_amd import engine
    W = torch.zeros(4, 64)
    with pytest.raises(ValueError, match="GPU"):
        engine.gather_rows(W, torch.zeros(2, dtype=torch.int64))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RechorusHipMissing):
        _lib.load()


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's table is the map from reference code to entry points: it has to mention all of them"""
    import re
    header = open(os.path.join(ROOT, "include", "rechorus_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", header)))
    missing = [n for n in names if n not in doc and not (n.endswith("_fwd") or n.endswith("_bwd")) ]
    # `rc_x_fwd/bwd` is written as one cell for pairs
    missing += [n for n in names if (n.endswith("_fwd") or n.endswith("_bwd")) and n not in doc and n[:-4] + "_fwd/bwd" not in doc]
    assert not missing, missing
