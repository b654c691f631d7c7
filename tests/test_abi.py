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
    # dropout: per-session k_r instead of the shared one, plus the kept dropout(pos_emb) [B, 2L, D], plus (round 6) the dropped
    # input rows [T, D] a first layer keeps when it applies the model's input dropout itself
    assert lib.t4r_xlnet_layer_ws_floats(1024, 20, 128, 4, 1) == n + (1024 - 1) * 2 * 20 * 128 + 1024 * 2 * 20 * 128 + T * 128
    assert lib.t4r_xlnet_layer_bwd_ws_floats(8, 20, 64, 4, 0) > 0
    assert lib.t4r_dropout_ctr_hi(3, 2, 4) == (3 << 16) | (2 << 8) | 4
    assert lib.t4r_xlnet_attn_bwd_ws_floats(8, 20, 64, 4) == 8 * (2 * 20 * 64 + 2 * 64)


def test_argument_errors_are_reported_not_crashes():
    lib = _lib.load()
    # d_head 300 is unsupported (the general kernels stop at 256): must come back as rc != 0 with a message, before any launch
    rc = lib.t4r_xlnet_attn_fwd(None, None, None, None, None, None, None, None, None, 1, 20, 4, 300, 0, 0.0, 0, 0, None)
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


# File-scope (namespace-scope) writable objects of the library: the header's "no global mutable state" convention,
# checked against the shared object's symbol table.  Every entry names WHY it may exist; none of them carries anything
# from one entry-point call to a later one except the documented switches.  A new file-scope variable fails this test
# until it is justified here (a process-wide forward->backward ring in head_split.hip is gone --
# that state now travels in the caller-owned t4r_head_note).
_ALLOWED_FILE_SCOPE_STATE = {
    "g_err": "thread-local error string (t4r_last_error)",
    "g_prec": "precision switch, set only through t4r_set_precision",
    "g_defer_join": "thread-local switch, set only through t4r_xlnet_layer_bwd_defer",
    "g_stack_prepared": "thread-local switch, set only through t4r_xlnet_stack_prepared",
    "g_side": "thread-local per-device side streams of the layer backward (created once, never freed)",
    "g_side_all": "registry of those side streams for t4r_xlnet_layer_bwd_join",
    "g_side_mu": "mutex of that registry",
    "g_sink": "thread-local, set and cleared INSIDE one t4r_xlnet_layer_bwd call (split-K partial sink)",
    "g_sg": "thread-local, set and cleared inside one t4r_gemm_softmax_grad_f32 call",
    "g_rank": "thread-local, set and cleared inside one t4r_rank_of_target_f32 call",
    "g_amax_a": "thread-local, set and cleared inside one layer call (operand maxima of a GEMM launch)",
    "g_amax_b": "same", "g_amax_n": "same", "g_amax_nb": "same",
    "g_cu_budget": "CUs the backward's token-tile kernels may count on, set only through t4r_xlnet_set_cu_budget (documented in "
                   "include/t4r_hip.h; per-row results do not depend on it, batch-reduced gradients are equal up to summation order)",
    "g_ff_amax": "thread-local, set and cleared inside one layer call",
    "g_ff_final_ctr": "thread-local, set and cleared inside one layer call (key of the fused model-level output dropout)",
    "g_ff_final_on": "same",
    "g_stack_gen": "thread-local, set by t4r_xlnet_stack_pos_dropout and consumed (cleared) by the next t4r_xlnet_stack_prepare",
    "g_w_amax_part": "thread-local, set by t4r_head_split_w_amax_hint and consumed (cleared) by the next t4r_head_split_logits_ce_dx",
    "g_w_amax_W": "same", "g_w_amax_n": "same",
    "g_ab_in_on": "thread-local, set and cleared inside one layer call (fused model-level input dropout of the first layer)",
    "g_ab_in_ctr": "same", "g_ab_hin": "same", "g_dh_in_on": "same", "g_dh_in_p": "same", "g_dh_in_seed": "same", "g_dh_in_ctr": "same",
    "g_red_side": "thread-local, set and cleared inside one layer call (reduction side stream)",
    "g_red_events": "same", "g_red_n": "same", "g_red_used": "same",
}


def test_no_file_scope_mutable_state_beyond_the_documented_list():
    import re
    import shutil
    import subprocess

    if not shutil.which("nm") or not shutil.which("c++filt"):
        import pytest
        pytest.skip("binutils not available")
    out = subprocess.run(f"nm --defined-only {_lib.LIB_PATH} | c++filt", shell=True, capture_output=True, text=True, check=True).stdout
    found = set()
    for line in out.splitlines():
        parts = line.split(None, 2)
        if len(parts) != 3 or parts[1] not in "bBdD":
            continue
        name = parts[2].strip()
        # function-local statics (lazily-set kernel attributes, env switches read once), compiler / HIP runtime objects,
        # kernel handles (device stubs demangle as functions) and header-library internals are not file-scope state of ours
        if "::" in name or "(" in name or name.startswith(("__", "_Z", "_DYNAMIC", "_GLOBAL_OFFSET_TABLE_", "DW.ref.", "std::", "guard variable", "vtable", "typeinfo")):
            continue
        found.add(re.sub(r"\s.*", "", name))
    assert "g_notes" not in found and "g_note_next" not in found
    extra = found - set(_ALLOWED_FILE_SCOPE_STATE)
    assert not extra, f"undocumented file-scope mutable state in libt4r_hip.so: {sorted(extra)}"


def test_head_note_is_caller_owned():
    lib = _lib.load()
    note = (ctypes.c_ulonglong * 8)()
    assert ctypes.sizeof(note) == 64
    assert lib.t4r_head_note_dw_form(ctypes.addressof(note)) == 0 and lib.t4r_head_note_dw_form(None) == 0
    hdr = open(_lib.HEADER_PATH).read()
    assert "typedef struct t4r_head_note { unsigned long long w[8]; } t4r_head_note;" in hdr


def test_product_build_reads_no_experiment_switch(monkeypatch):
    """round 6: A/B / tuning / fallback-forcing switches are compile-time (-DT4R_EXPERIMENTAL): the product library says so, the
    host side then ignores their names too (both sides of the ABI agree on which kernel families can run), and no C source calls
    getenv() outside the helper and the one documented product variable"""
    import glob
    import re

    from transformers4rec_amd import _lib

    assert _lib.load().t4r_experimental_build() == 0
    monkeypatch.setenv("T4R_XLNET_FUSED", "0")
    assert _lib.exp_env("T4R_XLNET_FUSED", "1") == "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    direct = []
    for f in glob.glob(os.path.join(root, "transformers4rec_amd", "csrc", "*")):
        for m in re.finditer(r'(?<![_a-z])getenv\("([A-Z0-9_]+)"\)', open(f).read()):
            direct.append((os.path.basename(f), m.group(1)))
    assert sorted(set(direct)) == [("gemm_f32.hip", "T4R_GEMM_PREC")], direct
    host = set()
    for f in glob.glob(os.path.join(root, "transformers4rec_amd", "*.py")):
        host |= set(re.findall(r'environ\.get\("(T4R_[A-Z0-9_]+)"', open(f).read()))
    assert host == {"T4R_HIP_LIB", "T4R_HEAD_MODE", "T4R_HEAD_AUTO_GB", "T4R_HEAD_CHUNK_MB", "T4R_HEAD_WS_GB"}, host
