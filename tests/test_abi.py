"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/t4r_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os

from transformers4rec_amd import _lib


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    syms = _lib.header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/t4r_hip.h but not exported"
    assert set(syms) == set(_lib._SIGS), "ctypes prototypes out of sync with the header"
    assert lib.t4r_abi_version() == 1


def test_no_torch_types_in_the_abi():
    import re

    decls = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    for word in ("at::", "torch", "Tensor", "std::", "c10"):
        assert word not in decls, f"{word} leaked into the C ABI"


def test_workspace_size_queries_are_pure():
    lib = _lib.load()
    n = lib.t4r_xlnet_layer_ws_floats(1024, 20, 128, 4, 0)
    T = 1024 * 20
    assert n >= T * 128 * (3 + 1 + 1 + 1 + 4 + 4 + 1)
    # dropout: per-session k_r instead of the shared one, plus the kept dropout(pos_emb) [B, 2L, D]
    assert lib.t4r_xlnet_layer_ws_floats(1024, 20, 128, 4, 1) == n + (1024 - 1) * 2 * 20 * 128 + 1024 * 2 * 20 * 128
    assert lib.t4r_xlnet_layer_bwd_ws_floats(8, 20, 64, 4, 0) > 0
    assert lib.t4r_dropout_ctr_hi(3, 2, 4) == (3 << 16) | (2 << 8) | 4
    assert lib.t4r_xlnet_attn_bwd_ws_floats(8, 20, 64, 4) == 8 * (2 * 20 * 64 + 2 * 64)


def test_argument_errors_are_reported_not_crashes():
    lib = _lib.load()
    # d_head 7 is unsupported: must come back as rc != 0 with a message, before any launch
    rc = lib.t4r_xlnet_attn_fwd(None, None, None, None, None, None, None, None, None, 1, 20, 4, 7, 0, 0.0, 0, 0, None)
    assert rc != 0 and b"d_head" in lib.t4r_last_error()
    rc = lib.t4r_mask_targets(None, None, 4, 0, 0, 0, None, None, None, 0.15, 0, 0, None, None, None)
    assert rc != 0 and b"L must be" in lib.t4r_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.T4RHipError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the extension is missing")
