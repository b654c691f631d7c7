"""ctypes binding of libt4r_hip.so (C ABI declared in include/t4r_hip.h).

There is NO fallback: if the shared library is missing or a call fails the product path
raises.  Build with `python -m transformers4rec_amd.build` (or __graft_entry__.build()).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# T4R_HIP_LIB selects another build of the same ABI (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("T4R_HIP_LIB") or os.path.join(_HERE, "lib", "libt4r_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "t4r_hip.h")

_P, _I, _L, _F, _Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_ulonglong
_C = {"p": _P, "i": _I, "l": _L, "f": _F, "Q": _Q}

# name -> (restype, argtype codes).  Must match include/t4r_hip.h (tests/test_abi.py checks the
# symbol list against the header).
_SIGS = {
    "t4r_abi_version": ("i", ""),
    "t4r_last_error": ("s", ""),
    "t4r_ragged_max_len": ("i", "ppip"),
    "t4r_ragged_to_padded": ("i", "ppppiii"),
    "t4r_ragged_gather_to_padded": ("i", "pppp" + "p" + "iii"),
    "t4r_seq_features_fwd": ("i", "pippppppiiiiiiipppp"),
    "t4r_embedding_bwd": ("i", "pppp" + "liiilii"),
    "t4r_sort_ids_ws_bytes": ("l", "l"),
    "t4r_sort_ids": ("i", "pp" + "lli" + "ppp" + "l"),
    "t4r_sort_ids_multi_ws_bytes": ("l", "li"),
    "t4r_sort_ids_multi": ("i", "ppilpp" + "ppp" + "l"),
    "t4r_embedding_bwd_sorted_ws_floats": ("l", "li"),
    "t4r_embedding_bwd_sorted": ("i", "ppppp" + "liiili" + "p"),
    "t4r_embedding_bag_fwd": ("i", "pp" + "li" + "pp" + "llii" + "p" + "li" + "p"),
    "t4r_embedding_bag_bwd_rows": ("i", "pp" + "lii" + "p" + "llii" + "p"),
    "t4r_apply_mask_fwd": ("i", "ppppiiii"),
    "t4r_apply_mask_bwd": ("i", "ppppiiii" + "p"),
    "t4r_apply_mask_bwd_ws_floats": ("l", "iii"),
    "t4r_apply_mask_bwd_to": ("i", "pppppiiii" + "p"),
    "t4r_mul": ("i", "ppppl"),
    "t4r_soft_embedding_fwd": ("i", "pppppppp" + "liif"),
    "t4r_soft_embedding_bwd": ("i", "pppppppppppp" + "liiiif" + "p"),
    "t4r_soft_embedding_bwd_ws_floats": ("l", "lii"),
    "t4r_mask_targets": ("i", "ppiiil" + "ppp" + "fQQ" + "ppp"),
    "t4r_compact_labels": ("i", "pppiil" + "pppp"),
    "t4r_gather_rows": ("i", "ppppii"),
    "t4r_scatter_rows_add": ("i", "ppppii"),
    "t4r_scatter_rows_dense": ("i", "pppippli"),
    "t4r_last_positions": ("i", "ppiiiilp"),
    "t4r_gemm_f32": ("i", "piiiiif" + "plplpl" + "pipl" + "iii" + "lll" + "fQQ"),
    "t4r_gemm_splitk_sink_begin": ("v", "pl"),
    "t4r_gemm_splitk_sink_flush": ("i", "p"),
    "t4r_gemm_splitk_sink_end": ("v", ""),
    "t4r_gemm_splitk_sink_bypassed": ("i", ""),
    "t4r_mha_fwd": ("i", "pppplplp" + "iiiii" + "fQQ" + "p"),
    "t4r_mha_bwd": ("i", "pppplpplppppl" + "iiiii" + "fQQ" + "p"),
    "t4r_add_pos_fwd": ("i", "ppppp" + "iii"),
    "t4r_add_pos_bwd": ("i", "ppp" + "iii"),
    "t4r_dropout_ctr_hi": ("Q", "Qii"),
    "t4r_dropout": ("i", "pppp" + "llfQQ"),
    "t4r_set_precision": ("v", "i"),
    "t4r_get_precision": ("i", ""),
    "t4r_gemm_softmax_grad_f32": ("i", "piiiif" + "plpppf" + "plpl" + "ii"),
    "t4r_set_tok_gemm_min_rows": ("v", "i"),
    "t4r_get_tok_gemm_min_rows": ("i", ""),
    "t4r_head_split_supported": ("i", "i"),
    "t4r_head_split_fwd_products": ("i", ""),
    "t4r_head_split_ws_bytes": ("l", "iii"),
    "t4r_head_split_prepare": ("i", "ppl" + "iii" + "p"),
    "t4r_head_split_fdx_supported": ("i", "i"),
    "t4r_head_split_logits_ce_dx": ("i", "pp" + "plpl" + "pl" + "pppp" + "plp" + "iii" + "ff" + "p"),
    "t4r_head_note_dw_form": ("i", "p"),
    "t4r_head_split_logits": ("i", "ppplpl" + "iiif" + "p"),
    "t4r_head_split_logits_ce": ("i", "ppplpl" + "pppp" + "iiiff" + "p"),
    "t4r_head_split_dw": ("i", "ppplpppf" + "pl" + "iiiii" + "fi" + "p"),
    "t4r_head_split_dx": ("i", "ppplpppf" + "plpl" + "iiiii" + "fi" + "p"),
    "t4r_head_split_recompute_supported": ("i", "i"),
    "t4r_head_split_prepare_rc": ("i", "ppl" + "iii" + "p"),
    "t4r_head_split_ce": ("i", "ppplpppp" + "iiiff" + "p"),
    "t4r_head_split_dw_rc": ("i", "ppplpppf" + "pl" + "iiifi" + "p"),
    "t4r_head_split_dx_rc": ("i", "ppplplpppf" + "pl" + "iiifi" + "p"),
    "t4r_add_layernorm_fwd": ("i", "pppppppp" + "iif" + "fQQ"),
    "t4r_add_layernorm_bwd": ("i", "pppppppppppp" + "iii" + "fQQ"),
    "t4r_colreduce_ws_floats": ("l", "li"),
    "t4r_act_bwd_bias": ("i", "pppppp" + "lii" + "fQQ"),
    "t4r_colsum": ("i", "pppp" + "lil"),
    "t4r_session_lengths": ("i", "pp" + "iiii" + "p"),
    "t4r_xlnet_attn_fwd": ("i", "ppppppppp" + "iiii" + "ifQQ" + "p"),
    "t4r_xlnet_attn_bwd_ws_floats": ("l", "iiii"),
    "t4r_xlnet_attn_bwd": ("i", "p" * 17 + "iiii" + "ifQQ" + "p"),
    "t4r_xlnet_fused_supported": ("i", "i"),
    "t4r_xlnet_fused_products": ("i", ""),
    "t4r_xlnet_ff_bwd_part_floats": ("l", "li"),
    "t4r_xlnet_layer_planes_floats": ("l", "i"),
    "t4r_xlnet_layer_prepare": ("i", "ppip"),
    "t4r_xlnet_qkv_proj": ("i", "pppp" + "li"),
    "t4r_xlnet_kr_proj": ("i", "pppp" + "li"),
    "t4r_xlnet_oproj_ln": ("i", "p" + "ppppp" + "pppp" + "liff" + "QQ"),
    "t4r_xlnet_ln1_bwd_part_floats": ("l", "li"),
    "t4r_xlnet_ln1_bwd": ("i", "p" + "ppppppp" + "pppppp" + "lif" + "QQ"),
    "t4r_xlnet_dh": ("i", "pppp" + "li"),
    "t4r_xlnet_set_cu_budget": ("v", "i"),
    "t4r_xlnet_get_cu_budget": ("i", ""),
    "t4r_device_cus": ("i", ""),
    "t4r_experimental_build": ("i", ""),
    "t4r_xlnet_attn_block_supported": ("i", "iii"),
    "t4r_xlnet_attn_block_fwd": ("i", "ppppp" + "l" + "pppp" + "ppppppp" + "iiii" + "ffQQQ" + "p"),
    "t4r_xlnet_layer_bwd_defer": ("v", "i"),
    "t4r_xlnet_layer_ws_offsets": ("i", "iiiii" + "pp"),
    "t4r_xlnet_stack_prepare": ("i", "ppiip" + "plp"),
    "t4r_xlnet_stack_pos_dropout": ("v", "fQQlp"),
    "t4r_xlnet_stack_prepared": ("v", "i"),
    "t4r_xlnet_layer_bwd_join": ("i", "p"),
    "t4r_xlnet_ff_planes_floats": ("l", "i"),
    "t4r_xlnet_ff_prepare": ("i", "ppppip"),
    "t4r_xlnet_ff_fwd": ("i", "p" + "pppppp" + "pppppp" + "iiff" + "QQQ"),
    "t4r_xlnet_ff_bwd": ("i", "p" + "pppppppp" + "pppppppp" + "iif" + "QQQ"),
    "t4r_xlnet_layer_ws_floats": ("l", "iiiii"),
    "t4r_xlnet_layer_bwd_ws_floats": ("l", "iiiii"),
    "t4r_xlnet_layer_fwd": ("i", "pppppp" + "iiiif" + "fQQi" + "pp"),
    "t4r_xlnet_layer_bwd": ("i", "ppppppppp" + "iiiif" + "fQQi" + "pp"),
    "t4r_softmax_ce_fwd": ("i", "pppppp" + "iilf"),
    "t4r_softmax_ce_bwd": ("i", "pppppp" + "iilf"),
    "t4r_linear_softmax_ce_chunk_floats": ("l", "ii"),
    "t4r_linear_softmax_ce_fwd": ("i", "pplpl" + "p" + "iiiffi" + "ppppp"),
    "t4r_linear_softmax_ce_bwd": ("i", "pplpl" + "ppp" + "iiiffi" + "pplpl"),
    "t4r_sampled_logits_fwd": ("i", "ppppppp" + "iiif" + "p"),
    "t4r_log_uniform_sample": ("i", "pp" + "ill" + "QQ"),
    "t4r_sampled_logits_bwd": ("i", "ppppppppp" + "iiif"),
    "t4r_sampled_logits_bwd_rows": ("i", "ppppppppp" + "iiif"),
    "t4r_topk": ("i", "pp" + "iili" + "pp"),
    "t4r_rank_of_target_f32": ("i", "p" + "iiif" + "pl" + "pl" + "ppp"),
    "t4r_swap_noise_ws_bytes": ("l", "l"),
    "t4r_swap_noise": ("i", "ppp" + "il" + "pllf" + "pp" + "QQ" + "pl"),
    "t4r_copy_cols": ("i", "pp" + "li" + "p" + "ili"),
    "t4r_seq_sum_cols": ("i", "pp" + "li" + "p" + "ili"),
    "t4r_adam_step": ("i", "ppppp" + "li" + "ffffff" + "i"),
    "t4r_adam_step_amax": ("i", "ppppp" + "li" + "ffffff" + "i" + "llp"),
    "t4r_head_split_w_amax_hint": ("v", "ppi"),
}

_lib = None


class T4RHipError(RuntimeError):
    pass


def header_symbols():
    """Function names declared in include/t4r_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(t4r_\w+)\s*\(", text)))


def load():
    """Loads the library and sets ctypes prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T4RHipError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python -m transformers4rec_amd.build`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_char_p if ret == "s" else (None if ret == "v" else _C[ret])
        fn.argtypes = [_C[a] for a in args]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().t4r_last_error()
        raise T4RHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)


def ptr_array(ptrs):
    """host array of device pointers (const T* const*)"""
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    return ctypes.cast(arr, ctypes.c_void_p), arr


def int_array(vals):
    arr = (ctypes.c_int * len(vals))(*vals)
    return ctypes.cast(arr, ctypes.c_void_p), arr


def long_array(vals):
    arr = (ctypes.c_long * len(vals))(*vals)
    return ctypes.cast(arr, ctypes.c_void_p), arr


def exp_env(name, default=None):
    """value of an EXPERIMENT switch: the environment variable `name` when the loaded library is an experiment build
    (-DT4R_EXPERIMENTAL, tools/experimental/build_variant.sh), else `default`.  The product build ignores these names on both
    sides of the C ABI (csrc/t4r_common.h: t4r_exp_getenv), so that the host never plans for a kernel family the library
    will not run.  The switches the product DOES read are listed in INTEGRATION.md section 5."""
    v = os.environ.get(name)
    if v is None:
        return default
    try:
        return v if load().t4r_experimental_build() else default
    except Exception:       # noqa: BLE001  (library not built yet: CPU-only import)
        return default
