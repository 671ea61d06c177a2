"""Tensor-level wrappers over the C-ABI (one function per entry point of include/c2mi.h).

All tensors live on the GPU; shapes/strides are checked here on the host before any
kernel is enqueued.  Nothing in this module computes on the CPU.
"""
import os

import torch

from . import _lib as L
from ._lib import check, dt_code, lib, ptr, stream


PROFILE_GEMM = None  # bench.py: list collecting (flops, start_event, end_event) of gene-width GEMM launches


class _GemmTimer:
    """Event pair on the launch stream around one gene-width GEMM (only while bench.py asks for it)."""

    def __init__(self, M, N, K):
        self.on = PROFILE_GEMM is not None and max(N, K) >= 2048 and min(M, N, K) >= 64  # weight is gene-width
        if self.on:
            self.flops = 2.0 * M * N * K
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            PROFILE_GEMM.append((self.flops, self.s, self.e))
        return False


def _gemm_code(dtype, x3):
    return L.F32X3 if (x3 and dtype == torch.float32) else dt_code(dtype)


def pitch2(n):
    """Row pitch (elements) of a 2-byte matrix with n columns that a gene-width GEMM reads: rows start on 256 bytes at gene
    widths, on 128 bytes at hidden widths, on 16 below 64 columns.  The LDS-DMA kernels move a row's 128-byte K-tile slab per
    eight lanes: from rows that start anywhere on 16 bytes (19 000 genes = 38 000 bytes apart) most slabs straddle two cache
    lines, and a reduction length that is not a whole number of 64-element K-tiles ends in a tile staged through registers.
    Measured on the C4 shapes (tools/dbg/gpitch.py, ktail.py): encoder forward 158.6 -> 136.8 us at a pitch of 19 072, the
    output layer's input gradient 134.5 -> 126.8, its forward at hidden width 820 179.7 -> 138.1 us at a pitch of 832."""
    n = int(n)
    if n >= 2048:
        return (n + 127) // 128 * 128
    if n >= 64:
        return (n + 63) // 64 * 64
    return (n + 7) // 8 * 8


def rows_pitch(G):
    """Row pitch (elements) of the resident counts [n, G] fp32 and mask [n, G] bytes: the next multiple of 128, so that every
    row starts on a 128-byte line in both (and on 16 / 4 bytes whatever G: the likelihood kernels' 16-byte row walk).  Pad
    columns are zero (count 0, mask 0).  NB loss at 4096 x 19 000: 236.6 -> 227.8 us (tools/dbg/nbpitch.py)."""
    return (int(G) + 127) // 128 * 128


def _row_major(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise L.C2Error(f"{name}: need a 2-D tensor with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    return t.stride(0)


def _operand(t, name, x3):
    """(2-D shape, leading dimension, dtype code) of a GEMM operand: a row-major matrix, or the two bf16 planes
    [2, rows, cols] (hi, lo) that split_bf16 makes of an fp32 matrix (dtype C2_BF16X3P).  The planes may be a view into
    padded storage (row pitch and plane stride of their own: the launch then names both planes by pointer)."""
    if t.dim() == 3:
        if t.dtype != torch.bfloat16 or t.shape[0] != 2 or (t.stride(2) != 1 and t.shape[2] > 1):
            raise L.C2Error(f"{name}: planes must be a [2, rows, cols] bf16 tensor with unit inner stride")
        return tuple(t.shape[1:]), t.stride(1), L.BF16X3P
    return tuple(t.shape), _row_major(t, name), _gemm_code(t.dtype, x3)


def _lo(t):
    """Pointer to the low plane of a planes operand."""
    return t.data_ptr() + t.stride(0) * 2


def split_bf16(src, out=None):
    """fp32 [rows, cols] -> bf16 planes [2, rows, cols]: hi = bf16(x), lo = bf16(x - hi).  `out` may be a view into padded
    storage (row pitch >= cols, any plane stride)."""
    L.require_gpu(src)
    rows, cols = src.shape
    assert src.dtype == torch.float32
    if out is None:
        out = torch.empty((2, rows, cols), dtype=torch.bfloat16, device=src.device)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (2, rows, cols) and (out.stride(2) == 1 or cols <= 1)
    check(lib.c2_split_bf16_planes(ptr(src), rows, cols, _row_major(src, "src"), ptr(out), _lo(out), out.stride(1), stream()))
    return out


def gemm_tile(dtype, M, N, R, x3=False, planes=False):
    """Block-tile edge the library uses for an [M,N] output reduced over R with operands of `dtype`
    (0: bf16 planes cannot take this shape)."""
    code = L.BF16X3P if planes else _gemm_code(dtype, x3)
    return int(lib.c2_gemm_tile(code, M, N, R))


def gemm_is_tiny(M, N, R):
    """An fp32-operand problem this small runs on the one-thread-per-output kernel (no planes, no tiles)."""
    return bool(lib.c2_gemm_is_tiny(M, N, R))


def planes_ok(M, N, R, lds=()):
    """Can an [M,N] = sum over R GEMM run on bf16 planes (LDS-DMA kernel)?  lds: the operands' leading dimensions."""
    return gemm_tile(None, M, N, R, planes=True) > 0 and all(int(x) % 8 == 0 for x in lds)


_SKINNY_SLABS = int(os.environ.get("C2_SKINNY_SLABS", "64"))


def pick_nslab(M, N, R, dtype=torch.float32, x3=False, max_slab=16, planes=False):
    """Number of slabs a reduction of length R is split into for an [M,N] output, so that the launch fills the
    256 CUs: >= ~2 blocks per CU of the register-staged 128x128 kernel, one block per CU of the LDS-DMA kernels."""
    if planes:
        R = max(64, (R + 7) // 8 * 8)  # (planes are reduced over their zero-padded length: a multiple of 8, one K-tile at least)
    tile = gemm_tile(dtype, M, N, R, x3, planes)
    tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
    bk = 64 if (planes or dtype != torch.float32) else 32
    target = 256 if (tile == 256 or planes) else 512
    if tiles >= target or R <= 4 * bk:
        return 1
    if tiles <= 4 and dtype == torch.bfloat16:
        max_slab = max(max_slab, _SKINNY_SLABS)  # a gene-width reduction into a one-batch-of-128 output: 4 tiles x 64 slabs fill the chip
    n = min(max_slab, (target + tiles - 1) // tiles, max(1, R // (4 * bk)))
    return max(1, n)


def prep_input(x, out, apply_log, center=None):
    """out = log(1 + x) or x, minus center [cols] per column when given (the centred encoder input of bf16 mode)."""
    L.require_gpu(x, out, center)
    if center is not None:
        assert center.dtype == torch.float32 and center.is_contiguous() and center.numel() >= x.shape[1]
    check(lib.c2_prep_input(ptr(x), x.shape[0], x.shape[1], _row_major(x, "x"), ptr(out), dt_code(out.dtype),
                            _row_major(out, "out"), int(apply_log), ptr(center), stream()))
    return out


def col_sum(x, apply_log, sums=None, ws=None):
    """[cols] float64: sum over the rows of log(1 + x) (or x), accumulated in double (order-independent: c2mi.h)."""
    L.require_gpu(x)
    rows, cols = x.shape
    assert x.dtype == torch.float32
    need = int(lib.c2_col_sum_ws(rows, cols))
    if sums is None:
        sums = torch.empty(cols, dtype=torch.float64, device=x.device)
    if ws is None:
        ws = torch.empty(need, dtype=torch.float64, device=x.device)
    assert sums.dtype == torch.float64 and sums.is_contiguous() and sums.numel() >= cols
    assert ws.dtype == torch.float64 and ws.numel() >= need
    check(lib.c2_col_sum(ptr(x), _row_major(x, "x"), rows, cols, int(apply_log), ptr(sums), ptr(ws), stream()))
    return sums


def col_sum_ws(rows, cols):
    return int(lib.c2_col_sum_ws(rows, cols))


def col_mean_from_sums(sums, cols, total_rows, out):
    L.require_gpu(sums, out)
    assert sums.dtype == torch.float64 and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= cols
    check(lib.c2_col_mean_from_sums(ptr(sums), cols, int(total_rows), ptr(out), stream()))
    return out


def col_mean(x, apply_log, out=None):
    """[cols] fp32: mean over the rows of log(1 + x) (or x) -- the centre of the encoder input (c2mi.h, c2_col_sum)."""
    rows, cols = x.shape
    assert rows > 0
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    return col_mean_from_sums(col_sum(x, apply_log), cols, rows, out)


def gemv_bias(w, c, bias, out):
    """out[n] = bias[n] + sum_k w[n, k] c[k] (fp32): the first layer's bias with the centre's contribution W c folded in."""
    L.require_gpu(w, c, out)
    N, K = w.shape
    assert w.dtype == torch.float32 and c.dtype == torch.float32 and out.dtype == torch.float32
    assert c.is_contiguous() and c.numel() >= K and out.is_contiguous() and out.numel() == N
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N)
    check(lib.c2_gemv_bias(ptr(w), _row_major(w, "w"), N, K, ptr(c), ptr(bias), ptr(out), stream()))
    return out


def rank1_add(g, r, c):
    """g[n, k] += r[n] * c[k] (fp32): the centre's term (sum_b dh_b) (x) c of the first layer's weight gradient."""
    L.require_gpu(g, r, c)
    N, K = g.shape
    assert g.dtype == torch.float32 and r.dtype == torch.float32 and c.dtype == torch.float32
    assert r.is_contiguous() and r.numel() == N and c.is_contiguous() and c.numel() >= K
    check(lib.c2_rank1_add(ptr(g), _row_major(g, "g"), N, K, ptr(r), ptr(c), stream()))
    return g


def linear_fwd(a, w, y, bias=None, nslab=1, x3=False):
    """y[s] = a @ w.T (+bias).  y: [M,N] or [nslab,M,N] fp32.  a / w: matrices of one dtype, or bf16 planes."""
    L.require_gpu(a, w, y)
    (M, K), lda, code = _operand(a, "a", x3)
    (N, K2), ldw, code_w = _operand(w, "w", x3)
    assert K2 == K and code == code_w and y.dtype == torch.float32
    y2 = y if y.dim() == 2 else y[0]
    assert tuple(y2.shape) == (M, N)
    ss = y.stride(0) if y.dim() == 3 else 0
    assert (y.dim() == 3 and y.shape[0] == nslab) or nslab == 1
    with _GemmTimer(M, N, K):
        if code == L.BF16X3P:
            check(lib.c2_linear_fwd_planes(ptr(a), _lo(a), lda, ptr(w), _lo(w), ldw, M, N, K, ptr(y), _row_major(y2, "y"),
                                           nslab, ss, ptr(bias), stream()))
        else:
            check(lib.c2_linear_fwd(ptr(a), lda, ptr(w), ldw, code, M, N, K, ptr(y), _row_major(y2, "y"), nslab, ss,
                                    ptr(bias), stream()))
    return y


def linear_fwd_f16out(a, w, y, bias=None, x3=False):
    """y = a @ w.T (+bias) rounded to IEEE halves (y: [M, N] float16): the decoder's logits in bf16 mode."""
    L.require_gpu(a, w, y)
    (M, K), lda, code = _operand(a, "a", x3)
    (N, K2), ldw, code_w = _operand(w, "w", x3)
    assert K2 == K and code == code_w and y.dtype == torch.float16 and tuple(y.shape) == (M, N)
    with _GemmTimer(M, N, K):
        check(lib.c2_linear_fwd_f16out(ptr(a), lda, ptr(w), ldw, code, M, N, K, ptr(y), _row_major(y, "y"), ptr(bias),
                                       stream()))
    return y


def nb_logits_f16_ok(B, G):
    return bool(lib.c2_recon_nb_logits_f16_ok(B, G))


def _rows_arg(a_full, rows, need, dtypes=(torch.bfloat16,)):
    assert a_full.dtype in dtypes and a_full.dim() == 2 and a_full.stride(1) == 1
    assert rows.dtype == torch.int32 and rows.is_contiguous() and rows.numel() >= need


def linear_fwd_rows(a_full, rows, M, w, y, bias=None, nslab=1):
    """y[s] = a_full[rows[:M]] @ w.T (+bias): the batch is read in place from the resident bf16 (or f16) matrix."""
    L.require_gpu(a_full, rows, w, y)
    _rows_arg(a_full, rows, M, (torch.bfloat16, torch.float16))
    N, K = w.shape
    assert w.dtype == a_full.dtype and a_full.shape[1] == K and y.dtype == torch.float32
    y2 = y if y.dim() == 2 else y[0]
    assert tuple(y2.shape) == (M, N) and ((y.dim() == 3 and y.shape[0] == nslab) or nslab == 1)
    ss = y.stride(0) if y.dim() == 3 else 0
    with _GemmTimer(M, N, K):
        check(lib.c2_linear_fwd_rows(ptr(a_full), a_full.stride(0), a_full.shape[0], ptr(rows), ptr(w), _row_major(w, "w"),
                                     dt_code(w.dtype), M, N, K, ptr(y), _row_major(y2, "y"), nslab, ss, ptr(bias), stream()))
    return y


def _rank1(rank1, N, K):
    """(row vector [N], column vector [K]) of the epilogue's rank-one term, as pointers (None, None when absent)."""
    if rank1 is None:
        return None, None
    r, c = rank1
    L.require_gpu(r, c)
    assert r.dtype == torch.float32 and c.dtype == torch.float32 and r.is_contiguous() and c.is_contiguous()
    assert r.numel() == N and c.numel() >= K
    return ptr(r), ptr(c)


def linear_wgrad_rows(dy, a_full, rows, dw, nslab=1, alpha=None, rank1=None):
    """dw[s] = alpha * dy.T @ a_full[rows[:M]]; rows must be readable up to M rounded up to 64."""
    L.require_gpu(dy, a_full, rows, dw)
    M, N = dy.shape
    _rows_arg(a_full, rows, (M + 63) // 64 * 64)
    K = a_full.shape[1]
    assert dy.dtype == torch.bfloat16 and dw.dtype == torch.float32
    d2 = dw if dw.dim() == 2 else dw[0]
    assert tuple(d2.shape) == (N, K) and ((dw.dim() == 3 and dw.shape[0] == nslab) or nslab == 1)
    ss = dw.stride(0) if dw.dim() == 3 else 0
    with _GemmTimer(M, N, K):
        check(lib.c2_linear_wgrad_rows(ptr(dy), _row_major(dy, "dy"), ptr(a_full), a_full.stride(0), a_full.shape[0],
                                       ptr(rows), M, N, K, ptr(dw), _row_major(d2, "dw"), nslab, ss, ptr(alpha),
                                       *_rank1(rank1, N, K), stream()))
    return dw


def linear_wgrad_rows_at(dyt, a_full, rows, dw, nslab=1, alpha=None, rank1=None):
    """linear_wgrad_rows with dy handed over transposed: dyt [N, M] bf16 (the cells contiguous)."""
    L.require_gpu(dyt, a_full, rows, dw)
    N, M = dyt.shape
    _rows_arg(a_full, rows, (M + 63) // 64 * 64)
    K = a_full.shape[1]
    assert dyt.dtype == torch.bfloat16 and dw.dtype == torch.float32
    d2 = dw if dw.dim() == 2 else dw[0]
    assert tuple(d2.shape) == (N, K) and ((dw.dim() == 3 and dw.shape[0] == nslab) or nslab == 1)
    ss = dw.stride(0) if dw.dim() == 3 else 0
    with _GemmTimer(M, N, K):
        check(lib.c2_linear_wgrad_rows_at(ptr(dyt), _row_major(dyt, "dyt"), ptr(a_full), a_full.stride(0),
                                          a_full.shape[0], ptr(rows), M, N, K, ptr(dw), _row_major(d2, "dw"), nslab, ss,
                                          ptr(alpha), *_rank1(rank1, N, K), stream()))
    return dw


def linear_wgrad_bt(dy, at, dw, nslab=1, alpha=None, rank1=None):
    """dw[s] = alpha * dy.T @ at.T with the second operand handed over transposed: dy [M, N], at [K, M], both bf16."""
    L.require_gpu(dy, at, dw)
    M, N = dy.shape
    K, M2 = at.shape
    assert M2 == M and dy.dtype == torch.bfloat16 and at.dtype == torch.bfloat16 and dw.dtype == torch.float32
    d2 = dw if dw.dim() == 2 else dw[0]
    assert tuple(d2.shape) == (N, K) and ((dw.dim() == 3 and dw.shape[0] == nslab) or nslab == 1)
    ss = dw.stride(0) if dw.dim() == 3 else 0
    with _GemmTimer(M, N, K):
        check(lib.c2_linear_wgrad_bt(ptr(dy), _row_major(dy, "dy"), ptr(at), _row_major(at, "at"), M, N, K, ptr(dw),
                                     _row_major(d2, "dw"), nslab, ss, ptr(alpha), *_rank1(rank1, N, K), stream()))
    return dw


def linear_dgrad(dy, w, da, nslab=1, alpha=None, x3=False):
    """da[s] = alpha * dy @ w.   dy [M,N], w [N,K], da [M,K] or [nslab,M,K]."""
    L.require_gpu(dy, w, da)
    (M, N), lddy, code = _operand(dy, "dy", x3)
    (N2, K), ldw, code_w = _operand(w, "w", x3)
    assert N2 == N and code == code_w and da.dtype == torch.float32
    d2 = da if da.dim() == 2 else da[0]
    assert tuple(d2.shape) == (M, K)
    ss = da.stride(0) if da.dim() == 3 else 0
    assert (da.dim() == 3 and da.shape[0] == nslab) or nslab == 1
    with _GemmTimer(M, N, K):
        if code == L.BF16X3P:
            check(lib.c2_linear_dgrad_planes(ptr(dy), _lo(dy), lddy, ptr(w), _lo(w), ldw, M, N, K, ptr(da),
                                             _row_major(d2, "da"), nslab, ss, ptr(alpha), stream()))
        else:
            check(lib.c2_linear_dgrad(ptr(dy), lddy, ptr(w), ldw, code, M, N, K, ptr(da), _row_major(d2, "da"), nslab, ss,
                                      ptr(alpha), stream()))
    return da


def linear_wgrad(dy, a, dw, nslab=1, alpha=None, x3=False, rank1=None):
    """dw[s] = alpha * dy.T @ a.   dy [M,N], a [M,K], dw [N,K] or [nslab,N,K]."""
    L.require_gpu(dy, a, dw)
    (M, N), lddy, code = _operand(dy, "dy", x3)
    (M2, K), lda, code_a = _operand(a, "a", x3)
    assert M2 == M and code == code_a and dw.dtype == torch.float32
    d2 = dw if dw.dim() == 2 else dw[0]
    assert tuple(d2.shape) == (N, K)
    ss = dw.stride(0) if dw.dim() == 3 else 0
    assert (dw.dim() == 3 and dw.shape[0] == nslab) or nslab == 1
    with _GemmTimer(M, N, K):
        if code == L.BF16X3P:
            check(lib.c2_linear_wgrad_planes(ptr(dy), _lo(dy), lddy, ptr(a), _lo(a), lda, M, N, K, ptr(dw),
                                             _row_major(d2, "dw"), nslab, ss, ptr(alpha), *_rank1(rank1, N, K), stream()))
        else:
            check(lib.c2_linear_wgrad(ptr(dy), lddy, ptr(a), lda, code, M, N, K, ptr(dw), _row_major(d2, "dw"), nslab, ss,
                                      ptr(alpha), *_rank1(rank1, N, K), stream()))
    return dw


def linear_bwd_tiny(dy, w, a, da, dw, alpha=None):
    """da = alpha * dy @ w and dw = alpha * dy.T @ a in one launch (fp32, both problems tiny: gemm_is_tiny).
    dy [M,N], w [N,K], a [M,K], da [M,K], dw [N,K]."""
    L.require_gpu(dy, w, a, da, dw)
    M, N = dy.shape
    K = w.shape[1]
    assert tuple(w.shape) == (N, K) and tuple(a.shape) == (M, K) and tuple(da.shape) == (M, K) and tuple(dw.shape) == (N, K)
    assert all(t.dtype == torch.float32 for t in (dy, w, a, da, dw))
    check(lib.c2_linear_bwd_tiny(ptr(dy), _row_major(dy, "dy"), ptr(w), _row_major(w, "w"), ptr(a), _row_major(a, "a"),
                                 M, N, K, ptr(da), _row_major(da, "da"), ptr(dw), _row_major(dw, "dw"), ptr(alpha), stream()))


def sum_slabs(slabs, out):
    n = out.numel()
    assert slabs.is_contiguous() and out.is_contiguous() and slabs.numel() == slabs.shape[0] * n
    check(lib.c2_sum_slabs(ptr(slabs), slabs.shape[0], n, ptr(out), n, stream()))
    return out


def sum_slabs_multi(jobs):
    """jobs: [(slabs [S, ...], out [...])]: every out = its slabs summed over S, one launch for all."""
    n = len(jobs)
    if n == 0:
        return
    arr = (L.SlabSum * n)()
    for i, (slabs, out) in enumerate(jobs):
        L.require_gpu(slabs, out)
        m = out.numel()
        assert slabs.is_contiguous() and out.is_contiguous() and slabs.numel() == slabs.shape[0] * m
        assert slabs.dtype == torch.float32 and out.dtype == torch.float32
        arr[i] = L.SlabSum(ptr(slabs), ptr(out), m, m, slabs.shape[0])
    check(lib.c2_sum_slabs_multi(arr, n, stream()))


def sum_slabs_2d(slabs, out):
    """out[r, c] = sum_s slabs[s, r, c] for a strided 2-D `out` (a block of a wider matrix)."""
    S, rows, cols = slabs.shape
    assert slabs.is_contiguous() and tuple(out.shape) == (rows, cols) and out.stride(1) == 1
    check(lib.c2_sum_slabs_2d(ptr(slabs), S, rows * cols, rows, cols, cols, ptr(out), out.stride(0), stream()))
    return out


def _slabs(y):
    """(ptr-tensor, ld, nslab, slab_stride, rows, cols) of a [M,N] or [S,M,N] fp32 tensor."""
    if y.dim() == 3:
        return y, y.stride(1), y.shape[0], y.stride(0), y.shape[1], y.shape[2]
    return y, y.stride(0), 1, 0, y.shape[0], y.shape[1]


def _planes_arg(planes, B, H):
    if planes is None:
        return None
    assert planes.dtype == torch.bfloat16 and tuple(planes.shape) == (2, B, H) and planes.is_contiguous()
    return ptr(planes)


def _planes_ld(planes, B, H):
    """(pointer, row pitch) of the planes a layer tail writes: dense [2, B, H], or a [2, B, H] view of padded storage whose
    low plane lies B x pitch elements behind the high one."""
    if planes is None:
        return None, 0
    assert planes.dtype == torch.bfloat16 and tuple(planes.shape) == (2, B, H) and (planes.stride(2) == 1 or H <= 1)
    ld = planes.stride(1)
    assert planes.stride(0) == B * ld, "the low plane must follow the high plane's B rows"
    return ptr(planes), ld


def epilogue_fwd(y, bias, cond, wc, mask, p_drop, use_ln, act, yhat=None, rstd=None, planes=None):
    L.require_gpu(y, act)
    yt, ldy, ns, ss, B, H = _slabs(y)
    assert y.dtype == torch.float32 and tuple(act.shape) == (B, H)
    C = 0
    if cond is not None and wc is not None:
        assert cond.dtype == torch.int64 and cond.is_contiguous() and cond.shape[0] == B
        C = cond.shape[1]
        assert tuple(wc.shape) == (H, C) and wc.dtype == torch.float32
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and tuple(mask.shape) == (B, H)
    if yhat is not None:
        assert yhat.is_contiguous() and tuple(yhat.shape) == (B, H) and yhat.dtype == torch.float32
    check(lib.c2_epilogue_fwd(ptr(yt), ldy, ns, ss, ptr(bias), ptr(cond) if C else None, C, ptr(wc) if C else None,
                              wc.stride(0) if C else 0, ptr(mask), float(p_drop), int(use_ln), B, H, ptr(act),
                              dt_code(act.dtype), _row_major(act, "act"), ptr(yhat), ptr(rstd),
                              *_planes_ld(planes, B, H), stream()))
    return act


def epilogue_bwd(da, mask, p_drop, use_ln, yhat, rstd, dh, extra=None, extra_scale=None, planes=None):
    L.require_gpu(da, dh)
    dt, ldda, ns, ss, B, H = _slabs(da)
    assert da.dtype == torch.float32 and tuple(dh.shape) == (B, H) and yhat.is_contiguous()
    if extra is not None:
        assert extra.dtype == torch.float32 and tuple(extra.shape) == (B, H)
    check(lib.c2_epilogue_bwd(ptr(dt), ldda, ns, ss, ptr(extra), extra.stride(0) if extra is not None else 0,
                              ptr(extra_scale), ptr(mask), float(p_drop), int(use_ln), ptr(yhat), ptr(rstd), B, H,
                              ptr(dh), dt_code(dh.dtype), _row_major(dh, "dh"), *_planes_ld(planes, B, H), stream()))
    return dh


def bn_ws(B, H):
    return int(lib.c2_bn_ws(B, H))


def _f32c(t, shape, name):
    assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(shape), name


def bn_stats(y, bias, cond, wc, h_out, part_out, ws):
    """Pre-norm h = slabs + bias + condition columns, and the local per-column (mean, M2) partial [2,H]."""
    L.require_gpu(y, h_out, part_out, ws)
    yt, ldy, ns, ss, B, H = _slabs(y)
    assert y.dtype == torch.float32
    _f32c(h_out, (B, H), "h_out")
    _f32c(part_out, (2, H), "part_out")
    assert ws.dtype == torch.float32 and ws.numel() >= bn_ws(B, H)
    C = 0
    if cond is not None and wc is not None:
        assert cond.dtype == torch.int64 and cond.is_contiguous() and cond.shape[0] == B
        C = cond.shape[1]
        assert tuple(wc.shape) == (H, C) and wc.dtype == torch.float32
    if bias is not None:
        _f32c(bias, (H,), "bias")
    check(lib.c2_bn_stats(ptr(yt), ldy, ns, ss, ptr(bias), ptr(cond) if C else None, C, ptr(wc) if C else None,
                          wc.stride(0) if C else 0, B, H, ptr(h_out), ptr(part_out), ptr(ws), stream()))
    return part_out


def bn_fwd(h, parts, rows_per_part, total_rows, gamma, beta, running_mean, running_var, num_batches_tracked,
           momentum, update_running, mask, p_drop, act, stats_out, planes=None, eps=1e-5):
    """parts: [n,2,H] partials (None: eval mode, the running statistics normalise).  h becomes yhat in place."""
    L.require_gpu(h, act, stats_out)
    B, H = h.shape
    _f32c(h, (B, H), "h")
    _f32c(gamma, (H,), "gamma")
    _f32c(beta, (H,), "beta")
    _f32c(stats_out, (2, H), "stats_out")
    n = 0
    if parts is not None:
        n = parts.shape[0]
        _f32c(parts, (n, 2, H), "parts")
    for t, nm in ((running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _f32c(t, (H,), nm)
    if num_batches_tracked is not None:
        assert num_batches_tracked.dtype == torch.int64 and num_batches_tracked.numel() == 1
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and tuple(mask.shape) == (B, H)
    assert tuple(act.shape) == (B, H)
    check(lib.c2_bn_fwd(ptr(h), ptr(parts), n, int(rows_per_part), int(total_rows), ptr(gamma), ptr(beta),
                        ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), float(momentum),
                        int(bool(update_running)), float(eps), ptr(mask), float(p_drop), B, H, ptr(act),
                        dt_code(act.dtype), _row_major(act, "act"), ptr(stats_out), _planes_arg(planes, B, H), stream()))
    return act


def bn_bwd_reduce(da, mask, p_drop, yhat, gamma, beta, dy_out, sums_out, dgamma, dbeta, ws, extra=None,
                  extra_scale=None):
    L.require_gpu(da, dy_out, sums_out, ws)
    dt, ldda, ns, ss, B, H = _slabs(da)
    assert da.dtype == torch.float32
    _f32c(yhat, (B, H), "yhat")
    _f32c(dy_out, (B, H), "dy_out")
    _f32c(sums_out, (2, H), "sums_out")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        if t is not None:
            _f32c(t, (H,), nm)
    assert ws.dtype == torch.float32 and ws.numel() >= bn_ws(B, H)
    if extra is not None:
        assert extra.dtype == torch.float32 and tuple(extra.shape) == (B, H)
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and tuple(mask.shape) == (B, H)
    check(lib.c2_bn_bwd_reduce(ptr(dt), ldda, ns, ss, ptr(extra), extra.stride(0) if extra is not None else 0,
                               ptr(extra_scale), ptr(mask), float(p_drop), ptr(yhat), ptr(gamma), ptr(beta), B, H,
                               ptr(dy_out), ptr(sums_out), ptr(dgamma), ptr(dbeta), ptr(ws), stream()))
    return sums_out


def bn_bwd(dy, yhat, sums, total_rows, batch_stats, gamma, stats, dh, planes=None):
    L.require_gpu(dy, dh)
    B, H = dy.shape
    _f32c(dy, (B, H), "dy")
    _f32c(yhat, (B, H), "yhat")
    _f32c(sums, (2, H), "sums")
    _f32c(stats, (2, H), "stats")
    _f32c(gamma, (H,), "gamma")
    assert tuple(dh.shape) == (B, H)
    check(lib.c2_bn_bwd(ptr(dy), ptr(yhat), ptr(sums), int(total_rows), int(bool(batch_stats)), ptr(gamma), ptr(stats),
                        B, H, ptr(dh), dt_code(dh.dtype), _row_major(dh, "dh"), _planes_arg(planes, B, H), stream()))
    return dh


def bias_cond_grad_ws(B, H, C):
    return int(lib.c2_bias_cond_grad_ws(B, H, C))


def bias_cond_grad(dh, cond, db, dwc, ws, alpha=None):
    L.require_gpu(dh, ws)
    B, H = dh.shape
    C = cond.shape[1] if (cond is not None and dwc is not None) else 0
    assert ws.numel() >= bias_cond_grad_ws(B, H, C)
    if dwc is not None:
        assert tuple(dwc.shape) == (H, C) and dwc.dtype == torch.float32
    check(lib.c2_bias_cond_grad(ptr(dh), dt_code(dh.dtype), _row_major(dh, "dh"), ptr(cond) if C else None, C, B, H,
                                ptr(db), ptr(dwc) if C else None, dwc.stride(0) if C else 0, ptr(alpha), ptr(ws), stream()))


def reparam_kl_fwd(mu, lv, eps, inv_B, use_own_kl, z32, zt, kl_out, ws):
    L.require_gpu(mu, lv, eps, z32)
    B, Ld = mu.shape
    assert mu.stride(0) == lv.stride(0) and eps.is_contiguous() and z32.is_contiguous()
    check(lib.c2_reparam_kl_fwd(ptr(mu), ptr(lv), ptr(eps), B, Ld, mu.stride(0), float(inv_B), int(use_own_kl), ptr(z32),
                                ptr(zt), dt_code(zt.dtype) if zt is not None else 0,
                                zt.stride(0) if zt is not None else 0, ptr(kl_out), ptr(ws), stream()))


def reparam_kl_bwd(dz, mu, lv, eps, inv_B, use_own_kl, g_kl, dmu, dlv, extra=None, extra_scale=None):
    L.require_gpu(dz, mu, dmu)
    dt, lddz, ns, ss, B, Ld = _slabs(dz)
    assert dmu.stride(0) == dlv.stride(0) and dmu.dtype == dlv.dtype
    check(lib.c2_reparam_kl_bwd(ptr(dt), lddz, ns, ss, ptr(extra), extra.stride(0) if extra is not None else 0,
                                ptr(extra_scale), ptr(mu), ptr(lv), ptr(eps), B, Ld, mu.stride(0), float(inv_B),
                                int(use_own_kl), ptr(g_kl), ptr(dmu), ptr(dlv), dt_code(dmu.dtype), dmu.stride(0),
                                stream()))


def recon_nb_ws(B, G, C):
    return int(lib.c2_recon_nb_ws(B, G, C))


def nb_row_const(x, goodloc, out=None):
    """out[b] = sum_g goodloc * lgamma(x+1) (data-only term of the NB likelihood)."""
    L.require_gpu(x, goodloc)
    B, G = x.shape
    assert goodloc.dtype in (torch.bool, torch.uint8) and tuple(goodloc.shape) == (B, G) and x.dtype == torch.float32
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=x.device)
    check(lib.c2_nb_row_const(ptr(x), _row_major(x, "x"), ptr(goodloc), _row_major(goodloc, "goodloc"), B, G, ptr(out),
                              stream()))
    return out


def recon_nb(logits, x, goodloc, sizefactor, theta, cond, inv_B, loss_out, dlogits, dtheta, ws, status, lgx1=None,
             row_index=None, dbias=None):
    """row_index: x / goodloc are the resident [N,G] matrices, batch row b = their row row_index[b].
    dbias: fp32 [G], receives the column sums of dlogits (bias gradient of the layer that made the logits)."""
    L.require_gpu(logits, x, goodloc, dlogits)
    B, G = logits.shape
    assert x.shape[1] == G and goodloc.shape == x.shape and (row_index is not None or x.shape[0] == B)
    if row_index is not None:
        assert row_index.dtype == torch.int64 and row_index.is_contiguous() and row_index.numel() == B and row_index.is_cuda
    assert logits.dtype == torch.float32 or (logits.dtype == torch.float16 and dbias is not None), \
        "half-precision logits go through the entry point that also forms the bias gradient"
    C = theta.shape[1]
    assert goodloc.dtype in (torch.bool, torch.uint8) and theta.is_contiguous() and cond.is_contiguous()
    assert cond.dtype == torch.int64 and tuple(cond.shape) == (B, C) and dtheta.is_contiguous()
    assert ws.numel() >= recon_nb_ws(B, G, C) and sizefactor.dtype == torch.float32 and sizefactor.is_contiguous()
    if lgx1 is not None:
        assert lgx1.dtype == torch.float32 and lgx1.is_contiguous() and lgx1.numel() == B and lgx1.is_cuda
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == G and dbias.is_cuda
        check(lib.c2_recon_nb_fwd_bwd_dbias(ptr(logits), dt_code(logits.dtype), _row_major(logits, "logits"), ptr(x),
                                            _row_major(x, "x"),
                                            ptr(goodloc), _row_major(goodloc, "goodloc"), ptr(sizefactor), ptr(theta),
                                            ptr(cond), C, B, G, float(inv_B), ptr(loss_out), ptr(dlogits),
                                            dt_code(dlogits.dtype), _row_major(dlogits, "dlogits"), ptr(dtheta), ptr(ws),
                                            ptr(status), ptr(lgx1), ptr(row_index), ptr(dbias), stream()))
        return
    check(lib.c2_recon_nb_fwd_bwd(ptr(logits), _row_major(logits, "logits"), ptr(x), _row_major(x, "x"), ptr(goodloc),
                                  _row_major(goodloc, "goodloc"), ptr(sizefactor), ptr(theta), ptr(cond), C, B, G,
                                  float(inv_B), ptr(loss_out), ptr(dlogits), dt_code(dlogits.dtype),
                                  _row_major(dlogits, "dlogits"), ptr(dtheta), ptr(ws), ptr(status), ptr(lgx1), ptr(row_index),
                                  stream()))


def recon_zinb(logits, pi, x, goodloc, sizefactor, theta, cond, inv_B, loss_out, dlogits, dpi, dtheta, ws, status,
               row_index=None, lgx1=None):
    L.require_gpu(logits, pi, x, goodloc, dlogits, dpi)
    B, G = logits.shape
    C = theta.shape[1]
    assert tuple(pi.shape) == (B, G) and dlogits.dtype == dpi.dtype and tuple(dpi.shape) == (B, G)
    assert goodloc.dtype in (torch.bool, torch.uint8) and theta.is_contiguous() and cond.is_contiguous()
    assert cond.dtype == torch.int64 and tuple(cond.shape) == (B, C) and dtheta.is_contiguous()
    assert ws.numel() >= recon_nb_ws(B, G, C) and sizefactor.dtype == torch.float32 and sizefactor.is_contiguous()
    assert x.shape[1] == G and goodloc.shape == x.shape and (row_index is not None or x.shape[0] == B)
    args = (ptr(logits), _row_major(logits, "logits"), ptr(pi), _row_major(pi, "pi"), ptr(x), _row_major(x, "x"),
            ptr(goodloc), _row_major(goodloc, "goodloc"), ptr(sizefactor), ptr(theta), ptr(cond), C, B, G, float(inv_B),
            ptr(loss_out), ptr(dlogits), ptr(dpi), dt_code(dlogits.dtype), _row_major(dlogits, "dlogits"),
            _row_major(dpi, "dpi"), ptr(dtheta), ptr(ws), ptr(status), ptr(row_index))
    if lgx1 is not None:  # the per-row sum of goodloc * lgamma(x + 1), evaluated once per dataset (nb_row_const)
        assert lgx1.dtype == torch.float32 and lgx1.is_contiguous() and lgx1.numel() == B
        check(lib.c2_recon_zinb_fwd_bwd_rc(*args, ptr(lgx1), stream()))
    else:
        check(lib.c2_recon_zinb_fwd_bwd(*args, stream()))


def recon_mse(logits, x, goodloc, inv_B, loss_out, dlogits, ws):
    L.require_gpu(logits, x, goodloc, dlogits)
    B, G = x.shape
    check(lib.c2_recon_mse_fwd_bwd(ptr(logits), _row_major(logits, "logits"), ptr(x), _row_major(x, "x"), ptr(goodloc),
                                   _row_major(goodloc, "goodloc"), B, G, float(inv_B), ptr(loss_out), ptr(dlogits),
                                   dt_code(dlogits.dtype), _row_major(dlogits, "dlogits"), ptr(ws), stream()))


def classifier_loss(u, classes, betaclass, loss_out, du, ws):
    L.require_gpu(u, classes, du)
    B, P = u.shape
    assert classes.dtype == torch.int64 and classes.is_contiguous() and tuple(classes.shape) == (B, P)
    check(lib.c2_classifier_loss_fwd_bwd(ptr(u), u.stride(0), ptr(classes), B, P, float(betaclass), ptr(loss_out), ptr(du),
                                         dt_code(du.dtype), du.stride(0), ptr(ws), stream()))


def partition(cond, n_conditions, main_col, boundary, part_id, wt, ws, status):
    L.require_gpu(cond, part_id, wt)
    B, C = cond.shape
    assert cond.dtype == torch.int64 and cond.is_contiguous() and part_id.dtype == torch.int32
    assert wt.numel() >= L.MMD_MAX_PARTS ** 2 and ws.numel() * ws.element_size() >= int(lib.c2_partition_ws(B))
    check(lib.c2_partition(ptr(cond), B, C, int(n_conditions), -1 if main_col is None else int(main_col),
                           -1 if boundary is None else int(boundary), ptr(part_id), ptr(wt), ptr(ws), ptr(status),
                           stream()))


def mmd_ws(Bl, Bg, D):
    return int(lib.c2_mmd_ws(Bl, Bg, D))


def mmd(feat, part_id, wt, row0, Bl, beta, loss_out, dfeat, ws):
    L.require_gpu(feat, part_id, wt, dfeat)
    Bg, D = feat.shape
    assert dfeat.is_contiguous() and tuple(dfeat.shape) == (Bl, D) and ws.numel() >= mmd_ws(Bl, Bg, D)
    check(lib.c2_mmd_fwd_bwd(ptr(feat), dt_code(feat.dtype), _row_major(feat, "feat"), ptr(part_id), ptr(wt), int(row0),
                             int(Bl), Bg, D, float(beta), ptr(loss_out), ptr(dfeat), D, ptr(ws), stream()))


def mmd_static_capacity():
    """Floats of the library's static MMD scratch array (include/c2mi.h: C2_MMD_STATIC_FLOATS)."""
    return int(lib.c2_mmd_static_floats())


def mmd_static(feat, part_id, wt, row0, Bl, beta, loss_out, dfeat):
    """mmd() on the library's file-scope scratch array: no workspace argument, no allocation of any kind.  Raises
    C2Error when mmd_ws(Bl, Bg, D) exceeds mmd_static_capacity().  Calls overlapping on one GPU race (shared array)."""
    L.require_gpu(feat, part_id, wt, dfeat)
    Bg, D = feat.shape
    assert dfeat.is_contiguous() and tuple(dfeat.shape) == (Bl, D)
    check(lib.c2_mmd_fwd_bwd_static(ptr(feat), dt_code(feat.dtype), _row_major(feat, "feat"), ptr(part_id), ptr(wt),
                                    int(row0), int(Bl), Bg, D, float(beta), ptr(loss_out), ptr(dfeat), D, stream()))


def adam_step(entries, lr, beta1, beta2, eps, weight_decay, clip_value, step_t):
    """entries: list of (p, g, m, v, shadow_or_None) contiguous fp32 tensors; shadow: a bf16 copy of p's shape, or the
    [2, *p.shape] bf16 planes (hi, lo) of a C2_BF16X3P operand."""
    arr, n = _adam_table(entries)
    check(lib.c2_adam_step(arr, n, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                           float(clip_value), int(step_t), stream()))


def _adam_table(entries):
    n = len(entries)
    arr = (L.AdamTensor * max(n, 1))()
    for i, (p, g, m, v, sh) in enumerate(entries):
        L.require_gpu(p, g, m, v)
        assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
        assert p.dtype == g.dtype == m.dtype == v.dtype == torch.float32 and g.numel() == p.numel()
        lo = None
        cols = ld = 0
        if isinstance(sh, tuple):  # (hi, lo) planes given separately (rows of a stacked operand, planes in padded storage)
            sh, lo = sh
            assert lo.dtype == torch.bfloat16 and lo.numel() == p.numel()
        f16 = 0
        if sh is not None:
            if sh.dtype == torch.float16:  # an IEEE-half operand copy (the encoder's first layer)
                assert lo is None
                f16 = 1
            else:
                assert sh.dtype == torch.bfloat16
                if lo is None and sh.numel() == 2 * p.numel() and sh.shape[0] == 2:
                    sh, lo = sh[0], sh[1]
            assert sh.numel() == p.numel()
            if not sh.is_contiguous():  # rows a pitch apart (zero-padded storage): the kernel maps its flat index
                assert sh.dim() == 2 and sh.stride(1) == 1 and sh.stride(0) % 4 == 0
                assert lo is None or (lo.shape == sh.shape and lo.stride() == sh.stride())
                cols, ld = sh.shape[1], sh.stride(0)
            else:
                assert lo is None or lo.is_contiguous()
        arr[i] = L.AdamTensor(ptr(p), ptr(g), ptr(m), ptr(v), ptr(sh), p.numel(), ptr(lo), f16, cols, ld)
    return arr, n


def adam_state(step, lr, beta1, beta2, eps, weight_decay, clip_value, device):
    """(state, hyper): the device-resident step state of c2_adam_prepare / c2_adam_step_dev (c2_adam_state: an int64
    step count followed by six doubles, held as 7 x 8 bytes) and the block its kernel constants are written to."""
    import numpy as np
    host = np.zeros(7, dtype=np.float64)
    host[1:] = (lr, beta1, beta2, eps, weight_decay, clip_value)
    host[:1].view(np.int64)[0] = int(step)
    state = torch.from_numpy(host).to(device)
    hyper = torch.zeros(int(lib.c2_adam_hyper_bytes()), dtype=torch.uint8, device=device)
    return state, hyper


def adam_state_step(state):
    """Host read (a sync) of the step count held in a device adam state."""
    return int(state[:1].view(torch.int64).item())


def adam_prepare(state, hyper):
    L.require_gpu(state, hyper)
    assert state.dtype == torch.float64 and state.numel() == 7 and state.is_contiguous()
    assert hyper.dtype == torch.uint8 and hyper.numel() >= int(lib.c2_adam_hyper_bytes())
    check(lib.c2_adam_prepare(ptr(state), ptr(hyper), stream()))


def adam_step_dev(entries, hyper):
    """adam_step with the step's constants in device memory (after adam_prepare)."""
    arr, n = _adam_table(entries)
    L.require_gpu(hyper)
    check(lib.c2_adam_step_dev(arr, n, ptr(hyper), stream()))


def transpose_bf16(src, dst):
    """dst [C, R] = src [R, C] transposed (bf16, unit inner strides)."""
    L.require_gpu(src, dst)
    R, Cc = src.shape
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16 and tuple(dst.shape) == (Cc, R)
    check(lib.c2_transpose_bf16(ptr(src), _row_major(src, "src"), R, Cc, ptr(dst), _row_major(dst, "dst"), stream()))
    return dst


def cast_bf16(src, dst):
    """fp32 -> bf16 (or IEEE half, by dst's dtype) copy of a contiguous tensor."""
    L.require_gpu(src, dst)
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    assert dst.dtype in (torch.bfloat16, torch.float16)
    fn = lib.c2_cast_bf16 if dst.dtype == torch.bfloat16 else lib.c2_cast_f16
    check(fn(ptr(src), ptr(dst), src.numel(), stream()))
    return dst


def gather_rows(X, goodloc, sizefactor, cond, classes, idx, x_out, goodloc_out, sf_out, cond_out, classes_out, xin_out,
                apply_log, rowconst=None, rowconst_out=None, idx32_out=None, center=None, xin2_out=None):
    L.require_gpu(X, idx, x_out, center, xin2_out)
    if xin2_out is not None:
        assert xin2_out.dtype == torch.bfloat16 and xin2_out.dim() == 2 and xin2_out.stride(1) == 1
    if center is not None:
        assert center.dtype == torch.float32 and center.is_contiguous() and center.numel() >= X.shape[1]
    if idx32_out is not None:
        assert idx32_out.dtype == torch.int32 and idx32_out.is_contiguous() and idx32_out.numel() >= idx.numel()
    G = X.shape[1]
    n = idx.numel()
    C = cond.shape[1] if cond is not None else 0
    P = classes.shape[1] if classes is not None else 0
    assert idx.dtype == torch.int64

    def pitch(t, what):  # rows `pitch` elements apart, unit stride inside a row (a dataset with padded rows: c2_gather_rows_ld)
        if t is None:
            return 0
        assert t.dim() == 2 and t.shape[1] == G and (t.stride(1) == 1 or t.shape[1] <= 1) and t.stride(0) >= G, what
        return t.stride(0)
    check(lib.c2_gather_rows_ld(ptr(X), pitch(X, "X"), ptr(goodloc), pitch(goodloc, "goodloc"), ptr(sizefactor), ptr(cond),
                                ptr(classes), G, C, P, ptr(idx), n, ptr(x_out), pitch(x_out, "x_out"), ptr(goodloc_out),
                                pitch(goodloc_out, "goodloc_out"), ptr(sf_out), ptr(cond_out), ptr(classes_out),
                                ptr(xin_out), dt_code(xin_out.dtype) if xin_out is not None else 0,
                                xin_out.stride(0) if xin_out is not None else 0, int(apply_log), ptr(rowconst),
                                ptr(rowconst_out), ptr(idx32_out), idx32_out.numel() if idx32_out is not None else 0,
                                ptr(center), ptr(xin2_out), xin2_out.stride(0) if xin2_out is not None else 0, stream()))


def defer_loss_sums(on):
    """While on, the loss entry points leave their second-stage sums to the next combine_losses[_dev] (one launch)."""
    return bool(lib.c2_defer_loss_sums(1 if on else 0))


def drop_pending_sums():
    lib.c2_drop_pending_sums()


def combine_losses(losses4, weights, out6):
    """out6 = (recon, kl, mmd, ce, weighted total, recon + kl + mmd)."""
    L.require_gpu(losses4, out6)
    assert losses4.dtype == torch.float32 and losses4.numel() == 4 and out6.dtype == torch.float32 and out6.numel() == 6
    w = [float(v) for v in weights]
    check(lib.c2_combine_losses(ptr(losses4), w[0], w[1], w[2], w[3], ptr(out6), stream()))
    return out6


def scale4(g_dev, weights, out4):
    """out4[i] = weights[i] * g (g: 0-d / 1-element fp32 device tensor, or None for 1)."""
    L.require_gpu(out4)
    assert out4.dtype == torch.float32 and out4.numel() == 4
    if g_dev is not None:
        assert g_dev.dtype == torch.float32 and g_dev.numel() == 1 and g_dev.is_cuda
    w = [float(v) for v in weights]
    check(lib.c2_scale4(ptr(g_dev), w[0], w[1], w[2], w[3], ptr(out4), stream()))
    return out4


def combine_losses_dev(losses4, w_dev, out6, accum6=None):
    """combine_losses with the four weights in device memory; accum6 (float64 [6]) += out6."""
    L.require_gpu(losses4, w_dev, out6)
    assert losses4.dtype == torch.float32 and losses4.numel() == 4 and out6.dtype == torch.float32 and out6.numel() == 6
    assert w_dev.dtype == torch.float32 and w_dev.numel() == 4 and w_dev.is_contiguous()
    if accum6 is not None:
        assert accum6.is_cuda and accum6.dtype == torch.float64 and accum6.numel() == 6 and accum6.is_contiguous()
    check(lib.c2_combine_losses_dev(ptr(losses4), ptr(w_dev), ptr(out6), ptr(accum6), stream()))
    return out6


def scale4_dev(g_dev, w_dev, out4):
    L.require_gpu(out4, w_dev)
    assert out4.dtype == torch.float32 and out4.numel() == 4
    assert w_dev.dtype == torch.float32 and w_dev.numel() == 4 and w_dev.is_contiguous()
    if g_dev is not None:
        assert g_dev.dtype == torch.float32 and g_dev.numel() == 1 and g_dev.is_cuda
    check(lib.c2_scale4_dev(ptr(g_dev), ptr(w_dev), ptr(out4), stream()))
    return out4


def scale(x, a_dev, y):
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(lib.c2_scale(ptr(x), ptr(a_dev), ptr(y), x.numel(), stream()))


def axpy(x, a_dev, y):
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(lib.c2_axpy(ptr(x), ptr(a_dev), ptr(y), x.numel(), stream()))


def scale4_dev_scale2(g_dev, w_dev, out4, jobs):
    """scale4_dev plus up to two vectors scaled by out4[0] in the same launch: jobs = [(x, y), ...] -> y = out4[0] * x."""
    L.require_gpu(w_dev, out4)
    assert w_dev.dtype == torch.float32 and w_dev.numel() == 4 and out4.dtype == torch.float32 and out4.numel() == 4
    assert len(jobs) <= 2
    args = []
    for x, y in list(jobs) + [(None, None)] * (2 - len(jobs)):
        if x is None:
            args += [None, None, 0]
            continue
        assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
        assert x.numel() == y.numel()
        args += [ptr(x), ptr(y), x.numel()]
    check(lib.c2_scale4_dev_scale2(ptr(g_dev), ptr(w_dev), ptr(out4), *args, stream()))
    return out4


def dropout_masks(out, p_drop, seed, offset):
    """Fill the flat uint8 buffer `out` with keep-masks (1 = keep)."""
    L.require_gpu(out)
    assert out.dtype == torch.uint8 and out.is_contiguous()
    check(lib.c2_dropout_masks(ptr(out), out.numel(), float(p_drop), int(seed) & (2 ** 64 - 1), int(offset), stream()))
    return out


def dropout_masks_dev(out, p_drop, seed, offset_dev, advance, lead=0):
    """dropout_masks with the counter offset in device memory (int64 [1], read as uint64): masks from *offset_dev + lead;
    the counter is advanced by `advance` after the masks are drawn."""
    L.require_gpu(out, offset_dev)
    assert out.dtype == torch.uint8 and out.is_contiguous()
    assert offset_dev.dtype == torch.int64 and offset_dev.numel() >= 1
    check(lib.c2_dropout_masks_dev(ptr(out), out.numel(), float(p_drop), int(seed) & (2 ** 64 - 1), ptr(offset_dev),
                                   int(lead), int(advance), stream()))
    return out


def noise_dev(masks, p_drop, eps, eps_first, seed, state, advance, lead=0):
    """One launch for a step's noise with the Philox counter in device memory (c2_noise_dev): keep-masks into the flat
    uint8 buffer `masks` (or None), standard normals into the fp32 tensor `eps` (or None; its elements are elements
    eps_first.. of the global batch's matrix), then state[0] += advance.  state: int64 [2] = (counter, 0)."""
    L.require_gpu(state)
    assert state.dtype == torch.int64 and state.numel() == 2 and state.is_contiguous()
    assert masks is None or (masks.dtype == torch.uint8 and masks.is_contiguous())
    assert eps is None or (eps.dtype == torch.float32 and eps.is_contiguous())
    check(lib.c2_noise_dev(ptr(masks) if masks is not None else None, masks.numel() if masks is not None else 0,
                           float(p_drop), ptr(eps) if eps is not None else None, eps.numel() if eps is not None else 0,
                           int(eps_first), int(seed) & (2 ** 64 - 1), ptr(state), int(lead), int(advance), stream()))


def graph_launch(graph_exec):
    """Launch an instantiated HIP graph (the integer handle torch.cuda.CUDAGraph.raw_cuda_graph_exec() returns) on the
    current stream."""
    check(lib.c2_graph_launch(int(graph_exec), stream()))


def decoder_output(logits, mode):
    """Softmax over the genes ("softmax") or ReLU ("relu") of the decoder's output layer, a new [B, G] fp32 tensor
    (reconstructLatent, celligner2.py:411-420)."""
    L.require_gpu(logits)
    assert logits.dtype == torch.float32 and logits.dim() == 2 and (logits.stride(1) == 1 or logits.numel() == 0)
    out = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
    check(lib.c2_decoder_output(ptr(logits), logits.stride(0), logits.shape[0], logits.shape[1],
                                {"softmax": 0, "relu": 1}[mode], ptr(out), out.stride(0), stream()))
    return out
