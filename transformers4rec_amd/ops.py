"""Thin functional layer over the C ABI (include/t4r_hip.h): tensor checks, output allocation
with the torch caching allocator, launch on torch's current HIP stream.  No arithmetic happens
here and nothing falls back to torch ops: a missing library raises (see _lib.load).
"""
import ctypes

import torch

from . import _lib
from ._lib import call, int_array, long_array, ptr_array

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_BIAS_RESID = 0, 1, 2, 3, 4
AGG = {"concat": 0, "element-wise-sum": 1, "element-wise-sum-item-multi": 2}
MASK_NONE, MASK_MLM, MASK_CLM, MASK_CLM_INFER = 0, 1, 2, 3
MLM_TRAIN, MLM_EVAL_LAST, MLM_EVAL_ALL, MLM_INFER, CLM_TRAIN, CLM_LAST, CLM_INFER = range(7)


_GPU_OK = False


def _stream():
    global _GPU_OK
    if not _GPU_OK:
        if not torch.cuda.is_available():
            raise _lib.T4RHipError("no GPU visible to torch: the HIP path cannot run and there is no CPU path")
        _GPU_OK = True
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise _lib.T4RHipError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _p(t, dtype=None, name="tensor"):
    return None if t is None else _chk(t, dtype, name)


_LD_ALIGN = int(_lib.exp_env("T4R_LOGITS_LD_ALIGN", "64"))


def pad_ld(V):
    """leading dimension used for [N, V] logits: rows start on a 256-byte boundary, so the 128-byte
    row segments the GEMM epilogue stores (and the CE / backward kernels read) never straddle a cache line"""
    return (V + _LD_ALIGN - 1) // _LD_ALIGN * _LD_ALIGN


# ------------------------------------------------------------------------------------ GEMM
PRECISIONS = {"fp32": 0, "fp32_bf16x3": 1, "bf16": 2, "fp16": 3, "auto": 4}


def set_precision(mode):
    """precision of every dense contraction launched from now on (include/t4r_hip.h: t4r_set_precision):
    "fp32" (fp32 matrix cores) | "fp32_bf16x3" (fp32-accurate on the bf16 cores, exact 3-way split) |
    "bf16" / "fp16" (mixed precision as the reference's AMP: half operands, fp32 accumulation, fp32 master
    weights) | "auto" (default: fp32 accuracy, the split form where it is faster).  Returns the previous mode."""
    prev = get_precision()
    _lib.load().t4r_set_precision(PRECISIONS[mode] if isinstance(mode, str) else int(mode))
    return prev


def get_precision():
    m = _lib.load().t4r_get_precision()
    return next(k for k, v in PRECISIONS.items() if v == m)


class precision:
    """with ops.precision("bf16"): ...  (restores the previous mode)"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_precision(self.mode)

    def __exit__(self, *exc):
        set_precision(self.prev)


def gemm(a, b, trans_a=False, trans_b=False, alpha=1.0, bias=None, epilogue=EPI_NONE, out=None,
         aux=None, splitk=1, accumulate=False, ldc=None, drop=(0.0, 0, 0)):
    """out[M,N] = alpha * op(a) @ op(b) (+epilogue).  a, b 2-D row-major fp32."""
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    Kb = b.shape[1] if trans_b else b.shape[0]
    if K != Kb:
        raise ValueError(f"gemm: inner dims differ ({K} vs {Kb})")
    lda, ldb = a.stride(0), b.stride(0)
    if a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError("gemm: operands must be row-major with unit inner stride")
    if out is None:
        ldc = N if ldc is None else ldc
        buf = torch.empty((M, ldc), device=a.device, dtype=torch.float32)
        out = buf[:, :N]
    ldc = out.stride(0)
    call("t4r_gemm_f32", _stream(), int(trans_a), int(trans_b), M, N, K, float(alpha),
         a.data_ptr(), lda, b.data_ptr(), ldb, out.data_ptr(), ldc,
         _p(bias, torch.float32, "bias"), int(epilogue),
         None if aux is None else aux.data_ptr(), 0 if aux is None else aux.stride(0),
         int(splitk), int(accumulate), 1, 0, 0, 0, float(drop[0]), int(drop[1]), int(drop[2]))
    return out


def gemm_wgrad(a, b, out):
    """out[M, N] += a[K, M]^T @ b[K, N] over a long K (a weight gradient: K = tokens) with DETERMINISTIC split-K: the partial
    tiles go to a scratch buffer and one more launch adds them in split order (csrc/gemm_f32.hip: the split-K sink) instead
    of fp32 atomics in arrival order.  Needs a dense `out`; otherwise (or if the partials do not fit) plain gemm(splitk=-1)."""
    M, N, K = a.shape[1], b.shape[1], a.shape[0]
    if not (out.is_contiguous() and (M * N) % 4 == 0 and out.data_ptr() % 16 == 0):
        return gemm(a, b, True, False, splitk=-1, accumulate=True, out=out)
    lib = _lib.load()
    splits = max(1, min(256, K // 320 + 1))      # the launcher's rule: >= 20 k-tiles of 16 per split, <= 256 splits
    ws = torch.empty(M * N * splits, device=a.device, dtype=torch.float32)
    lib.t4r_gemm_splitk_sink_begin(ws.data_ptr(), ws.numel())
    try:
        gemm(a, b, True, False, splitk=-1, accumulate=True, out=out)
        call("t4r_gemm_splitk_sink_flush", _stream())
        bypassed = lib.t4r_gemm_splitk_sink_bypassed()
    finally:
        lib.t4r_gemm_splitk_sink_end()
    if bypassed:       # the sizing above is a copy of the launcher's rule: if they ever part, say so instead of silently losing
        import warnings   # the bit-reproducibility this function exists for (the sum itself is still correct)

        warnings.warn(f"gemm_wgrad: {bypassed} split-K launch(es) found no room in the deterministic sink ({splits} splits "
                      f"planned for M={M} N={N} K={K}) and used fp32 atomics: this gradient is not bit-reproducible", RuntimeWarning)
    return out


def gemm_softmax_grad(logits, lse, labels, grad_out, V, b, trans_a, alpha=1.0, label_smoothing=0.0,
                      out=None, splitk=1, accumulate=False):
    """trans_a=False: out[N_rows, N] = alpha * dlogits @ b[V, N] ; True: out[V, N] = alpha * dlogits^T @ b[N_rows, N]
    with dlogits = softmax-CE gradient formed on the fly from (logits, lse, labels, grad_out)."""
    n_rows = logits.shape[0]
    N = b.shape[1]
    ld = logits.stride(0) if n_rows > 1 else max(logits.shape[1], V)
    if out is None:
        out = torch.empty((V if trans_a else n_rows, N), device=logits.device, dtype=torch.float32)
    call("t4r_gemm_softmax_grad_f32", _stream(), int(trans_a), n_rows, V, N, float(alpha), logits.data_ptr(), ld,
         _chk(lse, torch.float32), _chk(labels, torch.int64), _p(grad_out), float(label_smoothing),
         b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), int(splitk), int(accumulate))
    return out


class tok_gemm_min_rows:
    """context manager: row threshold of the token-stationary body GEMM (csrc/tok_gemm.hip; default 32 768)"""

    def __init__(self, rows):
        self.rows = int(rows)

    def __enter__(self):
        lib = _lib.load()
        self.prev = lib.t4r_get_tok_gemm_min_rows()
        lib.t4r_set_tok_gemm_min_rows(self.rows)
        return self

    def __exit__(self, *exc):
        _lib.load().t4r_set_tok_gemm_min_rows(self.prev)
        return False


# ---- materialised head for d_model <= 128 (csrc/head_split.hip)
def head_split_supported(D):
    return bool(_lib.load().t4r_head_split_supported(int(D)))


def head_split_ws_bytes(N, V, D):
    """bytes of workspace head_split_prepare allocates for these sizes (0: unsupported width)"""
    return int(_lib.load().t4r_head_split_ws_bytes(int(N), int(V), int(D)))


def head_split_prepare(x, V):
    """cuts the head's input rows x [N, D] into the plane blocks of the three products; returns the workspace"""
    N, D = x.shape
    nbytes = _lib.load().t4r_head_split_ws_bytes(N, int(V), D)
    ws = torch.empty(max(nbytes, 16), device=x.device, dtype=torch.uint8)
    call("t4r_head_split_prepare", _stream(), _chk(x, torch.float32), x.stride(0), N, D, int(V), ws.data_ptr())
    # the host note the forward product leaves for d X / d W (include/t4r_hip.h: t4r_head_note, 64 bytes) lives and dies with
    # the workspace: a CPU int64[8] tensor, so that it can also travel through the dispatcher as an operator output
    # (torch_ops.next_item_head) and be re-attached to the workspace tensor its backward operator receives
    ws.t4r_note = torch.zeros(8, dtype=torch.int64)
    return ws


def _note(ws):
    n = getattr(ws, "t4r_note", None)
    return None if n is None else n.data_ptr()


def head_split_dw_form(ws):
    """which kernel the last head_split_dw on this workspace ran: 0 none yet, 1 three bf16 planes, 2 two-way fp16 split
    with per-item scales"""
    return _lib.load().t4r_head_note_dw_form(_note(ws))


def head_split_logits(ws, x, W, alpha=1.0, ldc=None):
    N, D = x.shape
    V = W.shape[0]
    ldc = V if ldc is None else ldc
    buf = torch.empty((N, ldc), device=x.device, dtype=torch.float32)
    call("t4r_head_split_logits", _stream(), ws.data_ptr(), _chk(W, torch.float32), W.stride(0), buf.data_ptr(), ldc,
         N, V, D, float(alpha), _note(ws))
    return buf[:, :V]


def head_split_logits_ce(ws, x, W, labels, alpha=1.0, label_smoothing=0.0, ldc=None):
    """logits [N, V] (view of an [N, ldc] buffer) + (mean loss, loss rows, lse) in one pass over the vocabulary"""
    N, D = x.shape
    V = W.shape[0]
    ldc = V if ldc is None else ldc
    dev = x.device
    buf = torch.empty((N, ldc), device=dev, dtype=torch.float32)
    loss_rows = torch.empty(N, device=dev, dtype=torch.float32)
    lse = torch.empty(N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    call("t4r_head_split_logits_ce", _stream(), ws.data_ptr(), _chk(W, torch.float32), W.stride(0), buf.data_ptr(), ldc,
         _chk(labels, torch.int64), loss_rows.data_ptr(), lse.data_ptr(), loss.data_ptr(), N, V, D, float(alpha),
         float(label_smoothing), _note(ws))
    return buf[:, :V], loss, loss_rows, lse


def head_split_fdx_supported(D):
    return bool(_lib.load().t4r_head_split_fdx_supported(int(D)))


def head_split_logits_ce_dx(ws, x, W, labels, alpha=1.0, label_smoothing=0.0, ldc=None, w_amax=None):
    """the one-pass forward (csrc/head_split.hip: head_fwd_dx_kernel): logits [N, V] (view of an [N, ldc] buffer), mean loss,
    loss rows, lse AND dx_unit [N, D] = d (mean loss) / d x for an upstream gradient of 1 -- the backward is dx_unit * grad_out
    plus head_split_dw on the same workspace; the logits are never read for d X"""
    N, D = x.shape
    V = W.shape[0]
    ldc = V if ldc is None else ldc
    dev = x.device
    buf = torch.empty((N, ldc), device=dev, dtype=torch.float32)
    loss_rows = torch.empty(N, device=dev, dtype=torch.float32)
    lse = torch.empty(N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    dx = torch.empty((N, D), device=dev, dtype=torch.float32)
    wsum = None
    if label_smoothing > 0:
        wsum = torch.zeros(D, device=dev, dtype=torch.float32)
        colsum_(W, wsum)
    if w_amax is not None:      # (partials, n) of ops.w_amax_of(the table's Parameter): max |W| without a pass over W
        _lib.load().t4r_head_split_w_amax_hint(_chk(W, torch.float32), _chk(w_amax[0], torch.float32), int(w_amax[1]))
    call("t4r_head_split_logits_ce_dx", _stream(), ws.data_ptr(), _chk(x, torch.float32), x.stride(0), _chk(W, torch.float32),
         W.stride(0), buf.data_ptr(), ldc, _chk(labels, torch.int64), loss_rows.data_ptr(), lse.data_ptr(), loss.data_ptr(),
         dx.data_ptr(), dx.stride(0), _p(wsum), N, V, D, float(alpha), float(label_smoothing), _note(ws))
    return buf[:, :V], loss, loss_rows, lse, dx


def head_split_dw(ws, logits, lse, labels, grad_out, V, D, out, alpha=1.0, label_smoothing=0.0, accumulate=True, yoff=0):
    N, Vc = logits.shape
    call("t4r_head_split_dw", _stream(), ws.data_ptr(), logits.data_ptr(), logits.stride(0), _chk(lse, torch.float32),
         _chk(labels, torch.int64), _p(grad_out), float(label_smoothing), out.data_ptr(), out.stride(0), N, Vc, int(V),
         int(yoff), int(D), float(alpha), int(accumulate), _note(ws))
    return out


def head_split_dx(ws, logits, lse, labels, grad_out, V, W, alpha=1.0, label_smoothing=0.0, out=None, accumulate=False, yoff=0):
    N, Vc = logits.shape
    D = W.shape[1]
    if out is None:
        out = torch.empty((N, D), device=logits.device, dtype=torch.float32)
    call("t4r_head_split_dx", _stream(), ws.data_ptr(), logits.data_ptr(), logits.stride(0), _chk(lse, torch.float32),
         _chk(labels, torch.int64), _p(grad_out), float(label_smoothing), W.data_ptr(), W.stride(0), out.data_ptr(),
         out.stride(0), N, Vc, int(V), int(yoff), D, float(alpha), int(accumulate), _note(ws))
    return out


# ---- the recomputing form: no [N, V] tensor (csrc/head_split.hip: head_ce_stats_h / head_dw_rc / head_dx_rc kernels)
def head_split_recompute_supported(D):
    return bool(_lib.load().t4r_head_split_recompute_supported(int(D)))


def head_split_ce(ws, x, W, labels, alpha=1.0, label_smoothing=0.0):
    """mean loss, loss rows, lse of softmax(alpha x W^T) vs labels from the prepared workspace; nothing of size [N, V] is written"""
    N, D = x.shape
    V = W.shape[0]
    dev = x.device
    loss_rows = torch.empty(N, device=dev, dtype=torch.float32)
    lse = torch.empty(N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    # the image of x only this form's d W reads (kept out of head_split_prepare: the materialised head would pay a launch for it)
    call("t4r_head_split_prepare_rc", _stream(), _chk(x, torch.float32), x.stride(0), N, D, int(V), ws.data_ptr())
    call("t4r_head_split_ce", _stream(), ws.data_ptr(), _chk(W, torch.float32), W.stride(0), _chk(labels, torch.int64),
         loss_rows.data_ptr(), lse.data_ptr(), loss.data_ptr(), N, V, D, float(alpha), float(label_smoothing), _note(ws))
    return loss, loss_rows, lse


def head_split_dw_rc(ws, W, lse, labels, grad_out, out, alpha=1.0, label_smoothing=0.0, accumulate=True):
    N = lse.shape[0]
    V, D = W.shape
    call("t4r_head_split_dw_rc", _stream(), ws.data_ptr(), _chk(W, torch.float32), W.stride(0), _chk(lse, torch.float32),
         _chk(labels, torch.int64), _p(grad_out), float(label_smoothing), out.data_ptr(), out.stride(0), N, V, D, float(alpha),
         int(accumulate), _note(ws))
    return out


def head_split_dx_rc(ws, x, W, lse, labels, grad_out, alpha=1.0, label_smoothing=0.0, out=None, accumulate=False):
    N, D = x.shape
    V = W.shape[0]
    if out is None:
        out = torch.empty((N, D), device=x.device, dtype=torch.float32)
    call("t4r_head_split_dx_rc", _stream(), ws.data_ptr(), _chk(x, torch.float32), x.stride(0), _chk(W, torch.float32), W.stride(0),
         _chk(lse, torch.float32), _chk(labels, torch.int64), _p(grad_out), float(label_smoothing), out.data_ptr(), out.stride(0),
         N, V, D, float(alpha), int(accumulate), _note(ws))
    return out


# ------------------------------------------------------------------------------------ LN / act
SITE_INPUT, SITE_POS, SITE_PROB, SITE_ATTN_OUT, SITE_FF_ACT, SITE_FF_OUT, SITE_FINAL = range(7)
LAYER_FUSE_FINAL = 0x100     # flag in xlnet_layer_fwd / _bwd's layer_idx: this layer also applies the model's output dropout
LAYER_FUSE_INPUT = 0x200     # ... this (first) layer applies the model's input dropout: h is the undropped input
NO_DROP = (0.0, 0, 0)


def dropout_ctr_hi(offset, layer, site):
    return _lib.load().t4r_dropout_ctr_hi(int(offset), int(layer), int(site))


def dropout(x, p, seed, ctr_hi, n_total=None, want_mask=False, out=None):
    """out[i] = x[i % x.numel()] * keep(i)/(1-p), i < n_total (default x.numel())."""
    n_src = x.numel()
    n = n_src if n_total is None else n_total
    if out is None:
        out = torch.empty(n, device=x.device, dtype=torch.float32)
    mask = torch.empty(n, device=x.device, dtype=torch.uint8) if want_mask else None
    call("t4r_dropout", _stream(), _chk(x, torch.float32), out.data_ptr(), _p(mask), n, n_src, float(p),
         int(seed), int(ctr_hi))
    return (out, mask) if want_mask else out


def add_layernorm_fwd(a, b, gamma, beta, eps, drop=NO_DROP):
    rows, D = a.shape[0], a.shape[1]
    y = torch.empty_like(a)
    mean = torch.empty(rows, device=a.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    call("t4r_add_layernorm_fwd", _stream(), _chk(a, torch.float32), _p(b, torch.float32),
         _chk(gamma), _chk(beta), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, float(eps),
         float(drop[0]), int(drop[1]), int(drop[2]))
    return y, mean, rstd


def add_layernorm_bwd(a, b, gamma, mean, rstd, dy, dgamma, dbeta, dx=None, accumulate_dx=False,
                      drop=NO_DROP):
    """-> dx (gradient of b), or (dx, dxa) with dropout (dxa = gradient of the dropped operand a)."""
    rows, D = a.shape
    if dx is None:
        dx = torch.empty_like(a)
    dxa = torch.empty_like(a) if drop[0] > 0 else None
    ws = _colred_ws(rows, 2 * D, a.device)
    call("t4r_add_layernorm_bwd", _stream(), _chk(a), _p(b), _chk(gamma), _chk(mean), _chk(rstd),
         _chk(dy), dx.data_ptr(), _p(dxa), _p(dgamma), _p(dbeta), ws.data_ptr(), rows, D,
         int(accumulate_dx), float(drop[0]), int(drop[1]), int(drop[2]))
    return dx if dxa is None else (dx, dxa)


def _colred_ws(rows, ncols, device):
    return torch.empty(_lib.load().t4r_colreduce_ws_floats(rows, ncols), device=device, dtype=torch.float32)


def act_bwd_bias(dact, pre, dbias, mode, out=None, drop=NO_DROP):
    rows, N = dact.shape
    out = dact if out is None else out
    ws = None if dbias is None else _colred_ws(rows, N, dact.device)
    call("t4r_act_bwd_bias", _stream(), _chk(dact), _chk(pre), out.data_ptr(), _p(dbias), _p(ws), rows, N,
         mode, float(drop[0]), int(drop[1]), int(drop[2]))
    return out


def colsum_(x, out):
    ws = _colred_ws(x.shape[0], x.shape[1], x.device)
    call("t4r_colsum", _stream(), _chk(x), _chk(out), ws.data_ptr(), x.shape[0], x.shape[1], x.stride(0))
    return out


# ------------------------------------------------------------------------------------ input block
def ragged_to_padded(values, offsets, L):
    rows = offsets.numel() - 1
    out = torch.empty((rows, L), device=values.device, dtype=values.dtype)
    if values.dtype not in (torch.int64, torch.float32):
        raise TypeError("ragged_to_padded: int64 or float32 values")
    call("t4r_ragged_to_padded", _stream(), _chk(values), _chk(offsets, torch.int64), out.data_ptr(),
         rows, L, values.element_size())
    return out


def ragged_gather_to_padded(values, offsets, row_ids, L):
    """out[i, :] = the list of row row_ids[i] of the ragged column (values, offsets), zero-padded /
    truncated to L; offsets=None gathers a scalar column (returns [rows])."""
    rows = row_ids.numel()
    if values.dtype not in (torch.int64, torch.float32):
        raise TypeError("ragged_gather_to_padded: int64 or float32 values")
    scalar = offsets is None
    out = torch.empty((rows,) if scalar else (rows, L), device=values.device, dtype=values.dtype)
    call("t4r_ragged_gather_to_padded", _stream(), _chk(values), _p(offsets, torch.int64),
         _chk(row_ids, torch.int64), out.data_ptr(), rows, 1 if scalar else L, values.element_size())
    return out


def ragged_max_len(offsets):
    out = torch.empty(1, device=offsets.device, dtype=torch.int32)
    call("t4r_ragged_max_len", _stream(), _chk(offsets, torch.int64), offsets.numel() - 1, out.data_ptr())
    return out


def seq_features_fwd(feats, agg, B, L_in, L_out, W, item_feat=-1, mask_mode=MASK_NONE, mask=None,
                     masked_emb=None, err_flag=None):
    """feats: list of dicts(kind, input, table, dim, col, rows).  Returns out [B, L_out, W]."""
    n = len(feats)
    dev = feats[0]["input"].device
    out = torch.empty((B, L_out, W), device=dev, dtype=torch.float32)
    kinds, _k = int_array([f["kind"] for f in feats])
    inputs, _i = ptr_array([_chk(f["input"]) for f in feats])
    tables, _t = ptr_array([0 if f.get("table") is None else _chk(f["table"], torch.float32) for f in feats])
    dims, _d = int_array([f["dim"] for f in feats])
    cols, _c = int_array([f.get("col", 0) for f in feats])
    rows, _r = long_array([f.get("rows", 0) for f in feats])
    call("t4r_seq_features_fwd", _stream(), n, kinds, inputs, tables, dims, cols, rows, AGG[agg] if
         isinstance(agg, str) else agg, item_feat, B, L_in, L_out, W, mask_mode,
         _p(mask), _p(masked_emb), out.data_ptr(), _p(err_flag))
    return out


def embedding_bwd(dout, ids, d_table, col, dim, padding_idx=0):
    """ids [B, L] (sequence feature) or [B] (per-session context feature, gradient summed over L)"""
    W = dout.shape[-1]
    ntok = dout.numel() // W
    ids_div = ntok // ids.numel()
    call("t4r_embedding_bwd", _stream(), _chk(dout, torch.float32), _chk(ids, torch.int64),
         _chk(d_table, torch.float32), ntok, W, col, dim, d_table.shape[0], padding_idx, ids_div)


def sort_ids(ids, rows, padding_idx=0):
    """stable sort of the lookups by table row -> (keys_sorted int32 [n], perm int32 [n]); padding /
    out-of-range ids carry the key `rows` and come last.  Depends on the ids only (forward-pass work)."""
    ids = ids.contiguous().view(-1)
    n = ids.numel()
    keys = torch.empty(n, device=ids.device, dtype=torch.int32)
    perm = torch.empty(n, device=ids.device, dtype=torch.int32)
    nb = _lib.load().t4r_sort_ids_ws_bytes(n)
    ws = torch.empty(max(nb, 1), device=ids.device, dtype=torch.uint8)
    call("t4r_sort_ids", _stream(), _chk(ids, torch.int64), n, int(rows), int(padding_idx), keys.data_ptr(),
         perm.data_ptr(), ws.data_ptr(), nb)
    return keys, perm


def sort_ids_multi(ids_list, rows_list, padding_idx_list):
    """sort_ids for several tables in ONE device sort (csrc/embedding_sorted.hip: t4r_sort_ids_multi): every ids tensor has
    the same number of lookups n -> [(keys_sorted, perm)] per table, each exactly what sort_ids gives for it alone (views
    of two [F * n] buffers)."""
    F = len(ids_list)
    ids_list = [i.contiguous().view(-1) for i in ids_list]
    n = ids_list[0].numel()
    if F > 16 or any(i.numel() != n for i in ids_list):
        return [sort_ids(i, r, p) for i, r, p in zip(ids_list, rows_list, padding_idx_list)]
    dev = ids_list[0].device
    keys = torch.empty(F * n, device=dev, dtype=torch.int32)
    perm = torch.empty(F * n, device=dev, dtype=torch.int32)
    nb = _lib.load().t4r_sort_ids_multi_ws_bytes(n, F)
    ws = torch.empty(max(nb, 1), device=dev, dtype=torch.uint8)
    parr, _k0 = ptr_array([_chk(i, torch.int64) for i in ids_list])
    rows, _k1 = long_array([int(r) for r in rows_list])
    pads, _k2 = int_array([int(p) for p in padding_idx_list])
    call("t4r_sort_ids_multi", _stream(), parr, F, n, rows, pads, keys.data_ptr(), perm.data_ptr(), ws.data_ptr(), nb)
    return [(keys[f * n:(f + 1) * n], perm[f * n:(f + 1) * n]) for f in range(F)]


def embedding_bwd_sorted(dout, keys, perm, d_table, col, dim, ids_div=1):
    """deterministic d_table[id] += gradient rows (ascending lookup order, one owner per row, no atomics).
    dout [n * ids_div, W]; (keys, perm) from sort_ids."""
    W = dout.shape[-1]
    n = keys.numel()
    if dout.numel() != n * ids_div * W:
        raise ValueError("embedding_bwd_sorted: dout rows != lookups * ids_div")
    ws = torch.empty(max(1, _lib.load().t4r_embedding_bwd_sorted_ws_floats(n, dim)), device=dout.device,
                     dtype=torch.float32)
    call("t4r_embedding_bwd_sorted", _stream(), _chk(dout, torch.float32), _chk(keys, torch.int32),
         _chk(perm, torch.int32), _chk(d_table, torch.float32), n, W, col, dim, d_table.shape[0], int(ids_div),
         ws.data_ptr())


BAG_COMBINER = {"sum": 0, "mean": 1, "sqrtn": 2}


def embedding_bag_fwd(table, values, offsets=None, combiner="mean", out=None, col=0, err_flag=None):
    """torch.nn.EmbeddingBag(mode=combiner) without a padding index.  values: int64 [B, K] / [B] (matrix form,
    offsets None) or [n] with offsets int64 [B] (ragged form).  -> [B, dim] (or the columns [col, col + dim) of `out`)."""
    rows, dim = table.shape
    if offsets is None:
        v2 = values.view(values.shape[0], -1)
        n_bags, fixed_k = v2.shape[0], v2.shape[1]
    else:
        n_bags, fixed_k = offsets.numel(), 0
    if out is None:
        out = torch.empty((n_bags, dim), device=table.device, dtype=torch.float32)
    call("t4r_embedding_bag_fwd", _stream(), _chk(table, torch.float32), rows, dim, _chk(values, torch.int64),
         _p(offsets, torch.int64), n_bags, values.numel(), fixed_k, BAG_COMBINER[combiner], out.data_ptr(),
         out.stride(0), col, _p(err_flag))
    return out


def embedding_bag_bwd_rows(dout, n_values, dim, offsets=None, fixed_k=0, combiner="mean", col=0):
    """gradient row of every member lookup, [n_values, dim] in lookup order (see t4r_embedding_bag_bwd_rows)"""
    rows = torch.empty((n_values, dim), device=dout.device, dtype=torch.float32)
    n_bags = dout.shape[0]
    call("t4r_embedding_bag_bwd_rows", _stream(), _chk(dout, torch.float32), dout.stride(0), col, dim,
         _p(offsets, torch.int64), n_bags, n_values, fixed_k, BAG_COMBINER[combiner], rows.data_ptr())
    return rows


def swap_noise(x, item_ids, p, pad_token=0, bern=None, perm=None, seed=0, ctr_hi=0):
    """tr.StochasticSwapNoise.augment for one feature: x [B, L] or [B] (int64 ids / fp32 values);
    item_ids [B, L] gives the padding mask.  bern (uint8, x.shape) / perm (int64 [#non-pad]) inject
    the draws; None -> device Philox draws."""
    if x.dtype not in (torch.int64, torch.float32):
        raise TypeError("swap_noise: int64 ids or fp32 values")
    n = x.numel()
    out = torch.empty_like(x)
    stride = item_ids.shape[1] if x.ndim == item_ids.ndim - 1 else 1
    nb = _lib.load().t4r_swap_noise_ws_bytes(n)
    ws = torch.empty(max(nb, 1), device=x.device, dtype=torch.uint8)
    if bern is not None:
        bern = bern.to(torch.uint8).contiguous()
    call("t4r_swap_noise", _stream(), _chk(x), out.data_ptr(), x.element_size(), n,
         _chk(item_ids, torch.int64), int(pad_token), stride, float(p), _p(bern, torch.uint8),
         _p(perm, torch.int64), int(seed), int(ctr_hi), ws.data_ptr(), nb)
    return out


def copy_cols_out(wide2d, col, dim):
    """-> contiguous [rows, dim] copy of wide2d[:, col:col+dim]"""
    rows, ldw = wide2d.shape
    out = torch.empty((rows, dim), device=wide2d.device, dtype=torch.float32)
    call("t4r_copy_cols", _stream(), _chk(wide2d, torch.float32), ldw, col, out.data_ptr(), dim, rows, 0)
    return out


def seq_sum_cols(wide2d, col, dim, B, L):
    """-> [B, dim] = sum over the L tokens of each session of wide2d[:, col:col+dim]"""
    out = torch.empty((B, dim), device=wide2d.device, dtype=torch.float32)
    call("t4r_seq_sum_cols", _stream(), _chk(wide2d, torch.float32), wide2d.shape[1], col, out.data_ptr(), dim,
         B, L)
    return out


def apply_mask_fwd_(x, mask, memb, mode):
    B, L, H = x.shape
    call("t4r_apply_mask_fwd", _stream(), _chk(x), _chk(mask), _chk(memb), B, L, H, mode)
    return x


def apply_mask_bwd_(dy, mask, d_memb, mode):
    B, L, H = dy.shape
    ws = torch.empty(max(1, _lib.load().t4r_apply_mask_bwd_ws_floats(B, L, H)), device=dy.device, dtype=torch.float32)
    call("t4r_apply_mask_bwd", _stream(), _chk(dy), _chk(mask), _chk(d_memb), B, L, H, mode, ws.data_ptr())
    return dy


def apply_mask_bwd(dy, mask, d_memb, mode):
    """out of place: -> dx (dy untouched); d_memb accumulated"""
    B, L, H = dy.shape
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    ws = torch.empty(max(1, _lib.load().t4r_apply_mask_bwd_ws_floats(B, L, H)), device=dy.device, dtype=torch.float32)
    call("t4r_apply_mask_bwd_to", _stream(), _chk(dy), dx.data_ptr(), _chk(mask), _chk(d_memb), B, L, H, mode, ws.data_ptr())
    return dx


def mul(a, b):
    out = torch.empty_like(a)
    call("t4r_mul", _stream(), _chk(a), _chk(b), out.data_ptr(), a.numel())
    return out


def soft_embedding_fwd(x, proj_w, proj_b, table, ln_w, ln_b, eps=1e-5):
    K, D = table.shape
    out = torch.empty(x.shape + (D,), device=x.device, dtype=torch.float32)
    call("t4r_soft_embedding_fwd", _stream(), _chk(x, torch.float32), _chk(proj_w), _chk(proj_b),
         _chk(table), _p(ln_w), _p(ln_b), out.data_ptr(), x.numel(), K, D, float(eps))
    return out


def soft_embedding_bwd(dout, x, proj_w, proj_b, table, ln_w, d_proj_w, d_proj_b, d_table, d_ln_w,
                       d_ln_b, col, eps=1e-5):
    K, D = table.shape
    W = dout.shape[-1]
    ws = torch.empty(max(1, _lib.load().t4r_soft_embedding_bwd_ws_floats(x.numel(), K, D)), device=dout.device, dtype=torch.float32)
    call("t4r_soft_embedding_bwd", _stream(), _chk(dout), _chk(x), _chk(proj_w), _chk(proj_b),
         _chk(table), _p(ln_w), _chk(d_proj_w), _chk(d_proj_b), _chk(d_table), _p(d_ln_w), _p(d_ln_b),
         x.numel(), W, col, K, D, float(eps), ws.data_ptr())


# ------------------------------------------------------------------------------------ masking
def mask_targets(item_ids, mode, padding_idx=0, bern=None, j1=None, j2=None, p=0.15, seed=0,
                 offset=0, want_counts=True):
    B, L = item_ids.shape
    Lout = L + 1 if mode == MLM_INFER else L
    dev = item_ids.device
    mask = torch.empty((B, Lout), device=dev, dtype=torch.bool)
    labels = torch.empty((B, Lout), device=dev, dtype=torch.int64)
    counts = torch.empty(B, device=dev, dtype=torch.int32) if want_counts else None
    if bern is not None:
        bern = bern.to(torch.uint8) if bern.dtype != torch.uint8 else bern
    call("t4r_mask_targets", _stream(), _chk(item_ids, torch.int64), B, L, mode, padding_idx,
         _p(bern), _p(j1, torch.int64), _p(j2, torch.int64), float(p), int(seed), int(offset),
         mask.data_ptr(), labels.data_ptr(), _p(counts))
    return mask, labels, counts


def compact_labels(labels, counts, padding_idx=0):
    """-> (n_labels [1] int32 device, label_pos [B*L] int32, labels_compact [B*L] int64)"""
    B, L = labels.shape
    dev = labels.device
    row_off = torch.empty(B, device=dev, dtype=torch.int32)
    n = torch.empty(1, device=dev, dtype=torch.int32)
    pos = torch.empty(B * L, device=dev, dtype=torch.int32)
    lab = torch.empty(B * L, device=dev, dtype=torch.int64)
    call("t4r_compact_labels", _stream(), _chk(labels, torch.int64), _chk(counts, torch.int32), B, L,
         padding_idx, row_off.data_ptr(), n.data_ptr(), pos.data_ptr(), lab.data_ptr())
    return n, pos, lab


def gather_rows(x2d, pos, n):
    D = x2d.shape[1]
    out = torch.empty((n, D), device=x2d.device, dtype=torch.float32)
    call("t4r_gather_rows", _stream(), _chk(x2d, torch.float32), _chk(pos, torch.int32), out.data_ptr(), n, D)
    return out


def scatter_rows_add_(dout, pos, dx2d):
    n, D = dout.shape
    call("t4r_scatter_rows_add", _stream(), _chk(dout), _chk(pos, torch.int32), _chk(dx2d), n, D)
    return dx2d


def scatter_rows_dense(src, pos, n, scale, T):
    """[T, D] zeros with row pos[r] = scale * src[r] (r < n; pos ascending, scale a device scalar or None): one launch"""
    D = src.shape[1]
    dx = torch.empty((T, D), device=src.device, dtype=torch.float32)
    call("t4r_scatter_rows_dense", _stream(), _chk(src), _chk(pos, torch.int32), int(n), _p(scale), dx.data_ptr(), int(T), D)
    return dx


def last_positions(item_ids, Lgrid, is_mlm, padding_idx=0):
    B, L = item_ids.shape
    pos = torch.empty(B, device=item_ids.device, dtype=torch.int32)
    call("t4r_last_positions", _stream(), _chk(item_ids, torch.int64), B, L, Lgrid, int(is_mlm),
         padding_idx, pos.data_ptr())
    return pos


# ------------------------------------------------------------------------------------ attention / layer
def session_lengths(item_ids, padding_idx=0, extra=0):
    """int32 [B]: number of non-padding positions of every session (+ extra: the [MASK] slot of the MLM
    inference grid) -- the key_len of the opt-in attention padding mask"""
    B, L = item_ids.shape
    out = torch.empty(B, device=item_ids.device, dtype=torch.int32)
    call("t4r_session_lengths", _stream(), _chk(item_ids, torch.int64), B, L, int(padding_idx), int(extra),
         out.data_ptr())
    return out


def xlnet_attn_fwd(q, k, v, k_r, r_w_bias, r_r_bias, B, L, n_head, drop=NO_DROP, key_len=None):
    """k_r [2L, D] (shared) or [B*2L, D] (per session).  key_len int32 [B]: opt-in padding mask."""
    D = q.shape[-1]
    per_b = int(k_r.shape[0] == B * 2 * L and B > 1)
    out = torch.empty((B * L, D), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, n_head, L), device=q.device, dtype=torch.float32)
    call("t4r_xlnet_attn_fwd", _stream(), _chk(q), _chk(k), _chk(v), _chk(k_r), _chk(r_w_bias),
         _chk(r_r_bias), out.data_ptr(), lse.data_ptr(), B, L, n_head, D // n_head, per_b,
         float(drop[0]), int(drop[1]), int(drop[2]), _p(key_len, torch.int32))
    return out, lse


def xlnet_attn_bwd(q, k, v, k_r, r_w_bias, r_r_bias, out, lse, dout, d_rw, d_rr, B, L, n_head,
                   drop=NO_DROP, key_len=None):
    D = q.shape[-1]
    dev = q.device
    per_b = int(k_r.shape[0] == B * 2 * L and B > 1)
    # planes of one [3, T, D] buffer, as the layer holds them
    dq, dk, dv = torch.empty((3, B * L, D), device=dev, dtype=torch.float32).unbind(0)
    dkr = torch.empty_like(k_r)
    nws = _lib.load().t4r_xlnet_attn_bwd_ws_floats(B, L, D, n_head)
    ws = torch.empty(nws, device=dev, dtype=torch.float32)
    call("t4r_xlnet_attn_bwd", _stream(), _chk(q), _chk(k), _chk(v), _chk(k_r), _chk(r_w_bias),
         _chk(r_r_bias), _chk(out), _chk(lse), _chk(dout), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
         dkr.data_ptr(), _chk(d_rw), _chk(d_rr), ws.data_ptr(), B, L, n_head, D // n_head, per_b,
         float(drop[0]), int(drop[1]), int(drop[2]), _p(key_len, torch.int32))
    return dq, dk, dv, dkr


def mha_fwd(q, k, v, B, L, n_head, causal, drop=NO_DROP, key_len=None):
    """q,k,v: [B*L, D] views that may be column slices of one [B*L, 3D] buffer (row stride = ld)."""
    D = q.shape[1]
    ld = q.stride(0)
    assert k.stride(0) == ld and v.stride(0) == ld and q.stride(1) == 1
    out = torch.empty((B * L, D), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, n_head, L), device=q.device, dtype=torch.float32)
    call("t4r_mha_fwd", _stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), ld, out.data_ptr(), D,
         lse.data_ptr(), B, L, n_head, D // n_head, int(causal), float(drop[0]), int(drop[1]), int(drop[2]),
         _p(key_len, torch.int32))
    return out, lse


def mha_bwd(q, k, v, out, lse, dout, B, L, n_head, causal, drop=NO_DROP, fused_out=False, key_len=None):
    """-> dq, dk, dv.  fused_out: one [B*L, 3D] buffer (column blocks q|k|v), returned as the single tensor."""
    D = q.shape[1]
    ld = q.stride(0)
    dev = q.device
    if fused_out:
        buf = torch.empty((B * L, 3 * D), device=dev, dtype=torch.float32)
        dq, dk, dv, ldd = buf[:, :D], buf[:, D:2 * D], buf[:, 2 * D:], 3 * D
    else:
        dq, dk, dv = (torch.empty((B * L, D), device=dev, dtype=torch.float32) for _ in range(3))
        buf, ldd = None, D
    call("t4r_mha_bwd", _stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), ld, _chk(out), _chk(dout), D,
         _chk(lse), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ldd, B, L, n_head, D // n_head, int(causal),
         float(drop[0]), int(drop[1]), int(drop[2]), _p(key_len, torch.int32))
    return buf if fused_out else (dq, dk, dv)


def add_pos_fwd(x, pos, token_type=None):
    B, L, D = x.shape
    out = torch.empty_like(x)
    call("t4r_add_pos_fwd", _stream(), _chk(x, torch.float32), _chk(pos), _p(token_type), out.data_ptr(), B, L, D)
    return out


def add_pos_bwd_(dy, d_pos):
    B, L, D = dy.shape
    call("t4r_add_pos_bwd", _stream(), _chk(dy), _chk(d_pos), B, L, D)


XLNET_PARAM_ORDER = ("q", "k", "v", "o", "r", "r_w_bias", "r_r_bias", "ln1_w", "ln1_b", "w1", "b1",
                     "w2", "b2", "ln2_w", "ln2_b")


def xlnet_layer_ws_floats(B, L, D, n_head, dropout=False):
    return _lib.load().t4r_xlnet_layer_ws_floats(B, L, D, n_head, int(bool(dropout)))


def xlnet_layer_bwd_ws_floats(B, L, D, n_head, dropout=False):
    return _lib.load().t4r_xlnet_layer_bwd_ws_floats(B, L, D, n_head, int(bool(dropout)))


def xlnet_pos_emb_dropout(pos_emb, B, p, seed, offset):
    """dropout(pos_emb expanded over the batch) [B, 2L, D] with the key every layer uses (HF modeling_xlnet.py:1143
    drops it once per forward): pass the result as `pos_emb_b` to xlnet_layer_fwd / _bwd of all layers"""
    return dropout(pos_emb.contiguous().view(-1), p, seed, dropout_ctr_hi(offset, 255, SITE_POS), n_total=B * pos_emb.numel())


def xlnet_layer_fwd(h, pos_emb, params, B, L, n_head, eps, ws=None, drop_p=0.0, seed=0, offset=0,
                    layer_idx=0, key_len=None, pos_emb_b=None, stack_prepared=False):
    """h [B*L, D]; params: sequence of 15 tensors in XLNET_PARAM_ORDER.  -> (h_out, ws)
    stack_prepared: `ws` already holds this layer's weight planes and k_r (xlnet_stack_prepare)"""
    D = h.shape[-1]
    if ws is None:
        ws = torch.empty(xlnet_layer_ws_floats(B, L, D, n_head, drop_p > 0), device=h.device,
                         dtype=torch.float32)
    out = torch.empty_like(h)
    parr, _keep = ptr_array([_chk(p, torch.float32, "xlnet param") for p in params])
    lib = _lib.load()
    if stack_prepared:
        lib.t4r_xlnet_stack_prepared(1)
    try:
        call("t4r_xlnet_layer_fwd", _stream(), _chk(h, torch.float32), _chk(pos_emb, torch.float32), parr,
             ws.data_ptr(), out.data_ptr(), B, L, D, n_head, float(eps), float(drop_p), int(seed),
             int(offset), int(layer_idx), _p(key_len, torch.int32), _p(pos_emb_b, torch.float32))
    finally:
        if stack_prepared:
            lib.t4r_xlnet_stack_prepared(0)
    return out, ws


def xlnet_stack_prepare(params_all, B, L, n_head, pos, drop, pos_dropout=None):
    """The prologue of a whole XLNet stack in two launches (csrc/xlnet_fused_attn.hip: t4r_xlnet_stack_prepare): the
    weight planes of every layer and every layer's k_r = pos @ r.  params_all: per layer the 15 tensors in
    XLNET_PARAM_ORDER; pos: what the layers would project ([B * 2L, D] dropped positional encoding, or [2L, D]).
    -> one workspace per layer, to be passed to xlnet_layer_fwd(..., ws=, stack_prepared=True)."""
    import ctypes

    D = params_all[0][0].shape[0]
    dev = params_all[0][0].device
    n_fl = xlnet_layer_ws_floats(B, L, D, n_head, drop)
    po, ko = ctypes.c_long(0), ctypes.c_long(0)
    call("t4r_xlnet_layer_ws_offsets", B, L, D, n_head, int(bool(drop)), ctypes.addressof(po), ctypes.addressof(ko))
    if po.value < 0:
        raise ValueError("xlnet_stack_prepare: no fused kernels for this width")
    ws = [torch.empty(n_fl, device=dev, dtype=torch.float32) for _ in params_all]
    flat = [_chk(t, torch.float32, "xlnet param") for layer in params_all for t in layer]
    parr, _k0 = ptr_array(flat)
    n = len(params_all)
    planes, _k1 = ptr_array([w.data_ptr() + 4 * po.value for w in ws])
    kr, _k2 = ptr_array([w.data_ptr() + 4 * ko.value for w in ws])
    pos2 = pos.reshape(-1, D)
    rows = pos2.shape[0]
    if pos_dropout is not None:
        # (p, seed, offset, out): `pos` is the plain [2L, D] encoding; the projection kernel masks it per session with this
        # forward's pos_emb dropout and leaves the dropped rows in out [B * 2L, D] (what xlnet_pos_emb_dropout returned)
        pd, seed, offset, out = pos_dropout
        _lib.load().t4r_xlnet_stack_pos_dropout(float(pd), int(seed), dropout_ctr_hi(offset, 255, SITE_POS), rows, _chk(out, torch.float32))
        rows = B * rows
    call("t4r_xlnet_stack_prepare", _stream(), parr, n, D, planes, _chk(pos2, torch.float32), rows, kr)
    return ws


def xlnet_layer_bwd(h, pos_emb, params, grads, ws, dh_out, B, L, n_head, eps, bws=None, drop_p=0.0,
                    seed=0, offset=0, layer_idx=0, key_len=None, pos_emb_b=None, defer_join=None):
    """defer_join: a list -> the call returns without waiting for its weight-gradient streams (csrc/xlnet_layer.hip:
    deferred join) and appends every buffer those streams may still be using to the list; the caller keeps the list
    alive until it has called xlnet_layer_bwd_join()."""
    D = h.shape[-1]
    if bws is None:
        bws = torch.empty(xlnet_layer_bwd_ws_floats(B, L, D, n_head, drop_p > 0), device=h.device,
                          dtype=torch.float32)
    dh_in = torch.empty_like(h)
    parr, _k1 = ptr_array([_chk(p, torch.float32) for p in params])
    garr, _k2 = ptr_array([_chk(g, torch.float32) for g in grads])
    lib = _lib.load()
    if defer_join is not None:
        lib.t4r_xlnet_layer_bwd_defer(1)
    try:
        call("t4r_xlnet_layer_bwd", _stream(), _chk(h), _chk(pos_emb), parr, garr, _chk(ws),
             bws.data_ptr(), _chk(dh_out), dh_in.data_ptr(), B, L, D, n_head, float(eps), float(drop_p),
             int(seed), int(offset), int(layer_idx), _p(key_len, torch.int32), _p(pos_emb_b, torch.float32))
    finally:
        if defer_join is not None:
            lib.t4r_xlnet_layer_bwd_defer(0)
    if defer_join is not None:
        defer_join.append((h, pos_emb, list(params), list(grads), ws, bws, dh_out, key_len, pos_emb_b))
    return dh_in


def xlnet_layer_bwd_join():
    """the current stream waits for the weight-gradient streams of every deferred xlnet_layer_bwd call so far"""
    call("t4r_xlnet_layer_bwd_join", _stream())


# ------------------------------------------------------------------------------------ head
# ------------------------------------------------------------------------------------ fused XLNet layer pieces
def xlnet_fused_supported(D):
    return bool(_lib.load().t4r_xlnet_fused_supported(int(D)))


def xlnet_layer_prepare(params, D):
    """bf16 planes of a layer's nine weight matrices (csrc/xlnet_fused_attn.hip); params in XLNET_PARAM_ORDER"""
    planes = torch.empty(_lib.load().t4r_xlnet_layer_planes_floats(D), device=params[0].device, dtype=torch.float32)
    ptrs, _keep = ptr_array([_chk(t, torch.float32) for t in params])
    call("t4r_xlnet_layer_prepare", _stream(), ptrs, D, planes.data_ptr())
    return planes


def xlnet_qkv_proj(h, planes):
    T, D = h.shape
    qkv = torch.empty((3, T, D), device=h.device, dtype=torch.float32)
    call("t4r_xlnet_qkv_proj", _stream(), _chk(h, torch.float32), planes.data_ptr(), qkv.data_ptr(), T, D)
    return qkv


def xlnet_kr_proj(pos, planes):
    rows, D = pos.shape
    kr = torch.empty((rows, D), device=pos.device, dtype=torch.float32)
    call("t4r_xlnet_kr_proj", _stream(), _chk(pos, torch.float32), planes.data_ptr(), kr.data_ptr(), rows, D)
    return kr


def xlnet_oproj_ln(av, h, planes, gamma, beta, eps, drop=NO_DROP, train=True):
    T, D = av.shape
    h1 = torch.empty_like(av)
    ao = torch.empty_like(av) if train else None
    mean = torch.empty(T, device=av.device) if train else None
    rstd = torch.empty(T, device=av.device) if train else None
    call("t4r_xlnet_oproj_ln", _stream(), _chk(av, torch.float32), _chk(h, torch.float32), planes.data_ptr(), _chk(gamma),
         _chk(beta), _p(ao), _p(mean), _p(rstd), h1.data_ptr(), T, D, float(eps), float(drop[0]), int(drop[1]), int(drop[2]))
    return h1, ao, mean, rstd


def xlnet_ln1_bwd(dy, ao, h, mean, rstd, gamma, planes, d_gamma, d_beta, drop=NO_DROP):
    T, D = dy.shape
    dh, dao, dav = torch.empty_like(dy), torch.empty_like(dy), torch.empty_like(dy)
    part = torch.empty(max(1, _lib.load().t4r_xlnet_ln1_bwd_part_floats(T, D)), device=dy.device)
    call("t4r_xlnet_ln1_bwd", _stream(), _chk(dy), _chk(ao), _chk(h), _chk(mean), _chk(rstd), _chk(gamma), planes.data_ptr(),
         dh.data_ptr(), dao.data_ptr(), dav.data_ptr(), _chk(d_gamma), _chk(d_beta), part.data_ptr(), T, D, float(drop[0]),
         int(drop[1]), int(drop[2]))
    return dh, dao, dav


def xlnet_dh_(dqkv, planes, dh):
    _, T, D = dqkv.shape
    call("t4r_xlnet_dh", _stream(), _chk(dqkv, torch.float32), planes.data_ptr(), _chk(dh, torch.float32), T, D)
    return dh


_CU_BUDGET_SET = False


def xlnet_set_cu_budget(cus):
    """CUs the backward's token-tile kernels may plan for (0 = the whole chip); process-wide (csrc/xlnet_fused.hip)"""
    global _CU_BUDGET_SET
    _lib.load().t4r_xlnet_set_cu_budget(int(cus))
    _CU_BUDGET_SET = int(cus) > 0


def xlnet_clear_cu_budget():
    """no-op unless a budget is set: the safety net of a backward pass that raised before its reducer cleared it"""
    if _CU_BUDGET_SET:
        xlnet_set_cu_budget(0)


def device_cus():
    return int(_lib.load().t4r_device_cus())


def xlnet_get_cu_budget():
    return int(_lib.load().t4r_xlnet_get_cu_budget())


def xlnet_attn_block_supported(L, D, n_head):
    return bool(_lib.load().t4r_xlnet_attn_block_supported(int(L), int(D), int(n_head)))


def xlnet_attn_block_fwd(h, planes, o, kr, r_w_bias, r_r_bias, gamma, beta, B, L, n_head, eps, drop_p=0.0, seed=0, ctr_prob=0,
                         ctr_out=0, key_len=None, train=True):
    """the attention half of a layer in one launch (csrc/xlnet_attn_block.hip): h [B L, D] -> h1, and what the backward needs"""
    T, D = h.shape
    dev = h.device
    qkv = torch.empty((3, T, D), device=dev)
    av, h1 = torch.empty((T, D), device=dev), torch.empty((T, D), device=dev)
    lse = torch.empty((B, n_head, L), device=dev)
    ao = torch.empty((T, D), device=dev) if train else None
    mean = torch.empty(T, device=dev) if train else None
    rstd = torch.empty(T, device=dev) if train else None
    per_session = kr.shape[0] == B * 2 * L and B > 1
    call("t4r_xlnet_attn_block_fwd", _stream(), _chk(h, torch.float32), planes.data_ptr(), _chk(o, torch.float32), _chk(kr, torch.float32),
         2 * L * D if per_session else 0, _chk(r_w_bias), _chk(r_r_bias), _chk(gamma), _chk(beta), qkv.data_ptr(), av.data_ptr(),
         lse.data_ptr(), _p(ao), _p(mean), _p(rstd), h1.data_ptr(), B, L, D, n_head, float(eps), float(drop_p), int(seed),
         int(ctr_prob), int(ctr_out), _p(key_len, torch.int32))
    return h1, dict(qkv=qkv, av=av, lse=lse, ao=ao, mean=mean, rstd=rstd)


def xlnet_ff_fwd(h1, planes, b1, b2, gamma, beta, eps, drop_p=0.0, seed=0, ctr_act=0, ctr_out=0, train=True):
    T, D = h1.shape
    dev = h1.device
    hout = torch.empty_like(h1)
    saved = None
    if train:
        saved = dict(ffpre=torch.empty((T, 4 * D), device=dev), ffact=torch.empty((T, 4 * D), device=dev),
                     ffout=torch.empty((T, D), device=dev), mean=torch.empty(T, device=dev), rstd=torch.empty(T, device=dev))
    g = (lambda k: saved[k].data_ptr()) if train else (lambda k: None)
    call("t4r_xlnet_ff_fwd", _stream(), _chk(h1, torch.float32), planes.data_ptr(), _chk(b1), _chk(b2), _chk(gamma), _chk(beta),
         g("ffpre"), g("ffact"), g("ffout"), g("mean"), g("rstd"), hout.data_ptr(), T, D, float(eps), float(drop_p), int(seed),
         int(ctr_act), int(ctr_out))
    return hout, saved


def xlnet_ff_bwd(dy, h1, saved, gamma, planes, d_gamma, d_beta, d_b2, d_b1, drop_p=0.0, seed=0, ctr_act=0, ctr_out=0):
    T, D = dy.shape
    dh1, dffout = torch.empty_like(dy), torch.empty_like(dy)
    dpre = torch.empty((T, 4 * D), device=dy.device)
    part = torch.empty(max(1, _lib.load().t4r_xlnet_ff_bwd_part_floats(T, D)), device=dy.device)
    call("t4r_xlnet_ff_bwd", _stream(), _chk(dy), _chk(saved["ffout"]), _chk(h1), _chk(saved["mean"]), _chk(saved["rstd"]),
         _chk(gamma), _chk(saved["ffpre"]), planes.data_ptr(), dh1.data_ptr(), dffout.data_ptr(), dpre.data_ptr(),
         _chk(d_gamma), _chk(d_beta), _chk(d_b2), _chk(d_b1), part.data_ptr(), T, D, float(drop_p), int(seed), int(ctr_act),
         int(ctr_out))
    return dh1, dffout, dpre


def softmax_ce_fwd(logits, labels, V, label_smoothing=0.0):
    """logits [N, ld] view or buffer whose row stride is the leading dimension."""
    N = logits.shape[0]
    ld = logits.stride(0) if N > 1 else max(logits.shape[1], V)
    dev = logits.device
    loss_rows = torch.empty(N, device=dev, dtype=torch.float32)
    lse = torch.empty(N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    call("t4r_softmax_ce_fwd", _stream(), logits.data_ptr(), _chk(labels, torch.int64),
         loss_rows.data_ptr(), lse.data_ptr(), loss.data_ptr(), N, V, ld, float(label_smoothing))
    return loss, loss_rows, lse


def softmax_ce_bwd(logits, labels, lse, grad_out, V, label_smoothing=0.0):
    N = logits.shape[0]
    ld = logits.stride(0) if N > 1 else max(logits.shape[1], V)
    buf = torch.empty((N, ld), device=logits.device, dtype=torch.float32)
    call("t4r_softmax_ce_bwd", _stream(), logits.data_ptr(), _chk(labels, torch.int64), _chk(lse),
         _p(grad_out), buf.data_ptr(), N, V, ld, float(label_smoothing))
    return buf


_CHUNK_BYTES = int(__import__("os").environ.get("T4R_HEAD_CHUNK_MB", "96")) << 20


def head_chunk_cols(N, V):
    """columns per vocabulary chunk of the non-materialising head: the [N, chunk] logits buffer stays
    inside the 256 MB Infinity Cache (default 96 MB), a multiple of 1024 columns, at most V rounded up"""
    c = max(1024, (_CHUNK_BYTES // (4 * max(N, 1))) // 1024 * 1024)
    return int(min(c, (V + 1023) // 1024 * 1024))


def linear_softmax_ce_fwd(x, W, labels, alpha=1.0, label_smoothing=0.0, chunk_cols=None):
    """mean CE of softmax(alpha * x @ W^T) vs labels without an [N, V] tensor -> (loss, loss_rows, lse)"""
    N, D = x.shape
    V = W.shape[0]
    c = chunk_cols or head_chunk_cols(N, V)
    dev = x.device
    buf = torch.empty(_lib.load().t4r_linear_softmax_ce_chunk_floats(N, c), device=dev, dtype=torch.float32)
    stats = torch.empty(4 * N, device=dev, dtype=torch.float32)
    loss_rows = torch.empty(N, device=dev, dtype=torch.float32)
    lse = torch.empty(N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    call("t4r_linear_softmax_ce_fwd", _stream(), _chk(x, torch.float32), x.stride(0), W.data_ptr(), W.stride(0),
         _chk(labels, torch.int64), N, V, D, float(alpha), float(label_smoothing), c, buf.data_ptr(),
         stats.data_ptr(), loss_rows.data_ptr(), lse.data_ptr(), loss.data_ptr())
    return loss, loss_rows, lse


def linear_softmax_ce_bwd(x, W, labels, lse, grad_out, dW=None, alpha=1.0, label_smoothing=0.0, chunk_cols=None):
    """-> dx [N, D]; dW [V, D] is accumulated into when given"""
    N, D = x.shape
    V = W.shape[0]
    c = chunk_cols or head_chunk_cols(N, V)
    buf = torch.empty(_lib.load().t4r_linear_softmax_ce_chunk_floats(N, c), device=x.device, dtype=torch.float32)
    dx = torch.empty((N, D), device=x.device, dtype=torch.float32)
    call("t4r_linear_softmax_ce_bwd", _stream(), _chk(x, torch.float32), x.stride(0), W.data_ptr(), W.stride(0),
         _chk(labels, torch.int64), _chk(lse, torch.float32), _p(grad_out), N, V, D, float(alpha),
         float(label_smoothing), c, buf.data_ptr(), dx.data_ptr(), D,
         None if dW is None else _chk(dW, torch.float32), 0 if dW is None else dW.stride(0))
    return dx


def sampled_logits_fwd(x, labels, W, neg, dist, temperature=1.0):
    N, D = x.shape
    S = neg.numel()
    out = torch.empty((N, S + 1), device=x.device, dtype=torch.float32)
    ws = torch.empty(max(1, S * D), device=x.device, dtype=torch.float32)
    call("t4r_sampled_logits_fwd", _stream(), _chk(x), _chk(labels, torch.int64), _chk(W),
         _chk(neg, torch.int64), _chk(dist, torch.float32), out.data_ptr(), N, D, S, float(temperature),
         ws.data_ptr())
    return out


def log_uniform_sample(n, min_id, max_id, seed, ctr_hi, device):
    """n ids ~ LogUniformSampler's distribution over [min_id, max_id) (with replacement), int64 [n]"""
    out = torch.empty(n, device=device, dtype=torch.int64)
    call("t4r_log_uniform_sample", _stream(), out.data_ptr(), int(n), int(min_id), int(max_id), int(seed), int(ctr_hi))
    return out


def sampled_logits_bwd(dlogits, x, labels, W, neg, dW, temperature=1.0):
    """dlogits is modified in place (entries of accidental hits are zeroed)"""
    N, D = x.shape
    dx = torch.empty_like(x)
    ws = torch.empty(2 * neg.numel() * D, device=x.device, dtype=torch.float32)
    call("t4r_sampled_logits_bwd", _stream(), _chk(dlogits), _chk(x), _chk(labels, torch.int64),
         _chk(W), _chk(neg, torch.int64), dx.data_ptr(), _chk(dW), ws.data_ptr(), N, D, neg.numel(),
         float(temperature))
    return dx


def sampled_logits_bwd_rows(dlogits, x, labels, W, neg, temperature=1.0):
    """row-sparse form: -> (dx [N, D], ids [N + S] = labels ++ neg, rows [N + S, D] gradient rows of W[ids]);
    dlogits is modified in place (entries of accidental hits are zeroed)"""
    N, D = x.shape
    S = neg.numel()
    dx = torch.empty_like(x)
    rows = torch.empty((N + S, D), device=x.device, dtype=torch.float32)
    ws = torch.empty(2 * S * D, device=x.device, dtype=torch.float32)
    call("t4r_sampled_logits_bwd_rows", _stream(), _chk(dlogits), _chk(x), _chk(labels, torch.int64),
         _chk(W), _chk(neg, torch.int64), dx.data_ptr(), rows.data_ptr(), ws.data_ptr(), N, D, S,
         float(temperature))
    ids = torch.cat([labels, neg.to(labels.dtype)])        # index glue (N + S int64)
    return dx, ids, rows


def scatter_rows_sorted(d_table, ids, rows, padding_idx=-1):
    """d_table[ids[i]] += rows[i], deterministic (sort + segmented sum); padding_idx = -1: every id counts"""
    keys, perm = sort_ids(ids, d_table.shape[0], padding_idx)
    embedding_bwd_sorted(rows, keys, perm, d_table, 0, rows.shape[1], 1)


def topk(scores, k, V=None):
    N = scores.shape[0]
    V = scores.shape[1] if V is None else V
    ld = scores.stride(0) if N > 1 else scores.shape[1]
    vals = torch.empty((N, k), device=scores.device, dtype=torch.float32)
    idx = torch.empty((N, k), device=scores.device, dtype=torch.int64)
    call("t4r_topk", _stream(), scores.data_ptr(), N, V, ld, k, vals.data_ptr(), idx.data_ptr())
    return vals, idx


# ------------------------------------------------------------------------------------ optimizer
def rank_of_target(x, W, labels, alpha=1.0, chunk=1024):
    """0-based rank of labels[i] among alpha * x[i] @ W^T (ties -> lower index first), int32 [N];
    the [N, V] scores are never materialised."""
    N, D = x.shape
    V = W.shape[0]
    rank = torch.empty(N, device=x.device, dtype=torch.int32)
    labels = labels.contiguous()
    # Target scores through the SAME kernel (diagonal of x_c @ W[y_c]^T), not a separate row-dot product: an output
    # element's bits do not depend on its tile position, so `score == target` detects exact ties (duplicated item rows)
    # and "ties go to the lower index" holds bit for bit, as in the top-k of the materialised scores.  A row-dot kernel
    # was tried in round 3 (cheaper by ~10 us) and broke exactly that: its rounding differs from the product's, so a tie
    # with a duplicate row turns into an arbitrary </> (tests/test_kernels_gpu.py::test_rank_of_target_matches_topk_and_sort).
    for s0 in range(0, N, chunk):
        xc, yc = x[s0: s0 + chunk], labels[s0: s0 + chunk]
        n = xc.shape[0]
        pos = yc.to(torch.int32)
        wy = gather_rows(W, pos, n)
        with precision("fp32"):         # the rank epilogue always runs on the fp32 matrix cores: same arithmetic here
            tgt = gemm(xc, wy, False, True, alpha=alpha).diagonal().contiguous()
        call("t4r_rank_of_target_f32", _stream(), n, V, D, float(alpha), _chk(xc, torch.float32), xc.stride(0),
             _chk(W, torch.float32), W.stride(0), _chk(tgt, torch.float32), _chk(yc, torch.int64),
             rank[s0: s0 + chunk].data_ptr())
    return rank


def adam_step_amax_(param, grad, exp_avg, exp_avg_sq, step, lo, hi, part, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                    weight_decay=0.0, grad_scale=1.0, zero_grad=True):
    """adam_step_ that also leaves the per-workgroup maxima of |param[lo:hi]| after the update in part[:n]; -> n"""
    lib = _lib.load()
    n = lib.t4r_adam_step_amax(_stream(), _chk(param, torch.float32), _chk(grad, torch.float32), _chk(exp_avg),
                               _chk(exp_avg_sq), param.numel(), int(step), float(lr), float(betas[0]), float(betas[1]),
                               float(eps), float(weight_decay), float(grad_scale), int(zero_grad), int(lo), int(hi),
                               _chk(part, torch.float32))
    if n < 0:
        raise _lib.T4RHipError(lib.t4r_last_error().decode())
    return n


def w_amax_of(W):
    """(partials, n) the optimizer left for parameter W (optim.FusedAdam: the maximum of the tied item table comes out of the
    Adam launch) if NOTHING has written W since -- same storage, same version counters -- else None"""
    rec = getattr(W, "_t4r_w_amax", None)
    if rec is None:
        return None
    part, n, ptr, v_param, flat, v_flat = rec
    if W.data_ptr() != ptr or W._version != v_param or flat._version != v_flat or not W.is_contiguous():
        return None
    return part, n


def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=0.0, grad_scale=1.0, zero_grad=True):
    call("t4r_adam_step", _stream(), _chk(param, torch.float32), _chk(grad, torch.float32),
         _chk(exp_avg), _chk(exp_avg_sq), param.numel(), int(step), float(lr), float(betas[0]),
         float(betas[1]), float(eps), float(weight_decay), float(grad_scale), int(zero_grad))
