"""Tensor-level wrappers over the C ABI (include/pianobart_hip.h). PyTorch is used only to own
device memory and the HIP stream; every op below launches a hand-written gfx950 kernel from
libpianobart_hip.so on torch's current stream. There is no fallback path."""
import ctypes
import os

import torch

from ._lib import (LIB, PB_BF16, PB_F32, PB_F32X3, GemmDesc, PBError, GEMM_ACCUM, GEMM_C_F32, GEMM_GELU,
                   GEMM_MUL_GELU_GRAD, GEMM_ROWDOT)

SEG_SIZES = [262, 134, 135, 262, 134, 38, 260, 55]          # PianoBart.classes order
SEG_OFF = [0]
for _n in SEG_SIZES:
    SEG_OFF.append(SEG_OFF[-1] + _n)
VOCAB = SEG_OFF[-1]                                         # 1280
_SEG9 = (ctypes.c_int32 * 9)(*SEG_OFF)
# the projected Octuple table keeps every stream in a fixed 264-row slot (max vocabulary 262, padded to a multiple of 8) so the
# per-stream table GEMMs are ONE batched launch with uniform strides
TAB_ROWS = 264
TAB_OFF = [TAB_ROWS * i for i in range(9)]
TAB_TOTAL = TAB_OFF[8]                                      # 2112
_TAB9 = (ctypes.c_int32 * 9)(*TAB_OFF)


def seg_array(offsets):
    return (ctypes.c_int32 * len(offsets))(*offsets)


def dtype_code(t):
    if t == torch.float32:
        return PB_F32
    if t == torch.bfloat16:
        return PB_BF16
    raise PBError('unsupported storage dtype %s' % t)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise PBError('pianobart_amd ops need HIP device tensors (got %s); there is no CPU path' % t.device)
    return ctypes.c_void_p(t.data_ptr())


# developer aid for same-box A/B runs (tools/ab_step.sh): extra flag bits for every GEMM, e.g. PB_GEMM_FLAGS=4096 = ordinary grids
_ENV_GEMM_FLAGS = int(os.environ.get('PB_GEMM_FLAGS', '0'))


def gemm(A, B, C, *, M, N, K, dtype, a_kc=True, b_kc=True, lda=None, ldb=None, ldc=None, bias=None, alpha=1.0,
         accum=False, c_f32=False, gelu_aux_out=None, gelu_grad_aux_in=None, ldaux=0, nb1=1, nb2=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), a_off=0, b_off=0, c_off=0, splitk=1, slabs=None, force_v1=False, tile128=False, tile256=False, dbg=0,
         colsum_out=None, colsum_ws=None, rowdot=None):
    """C[m,n] (+)= epi(alpha * sum_k A(m,k) B(n,k)). a_off/b_off/c_off are element offsets into the tensors.
    rowdot = (aux (M, N) storage dtype, out (N / 64, ld) f32, ld): PB_GEMM_ROWDOT, out[n / 64][m] = sum over the 64-column group of C * aux."""
    d = GemmDesc()
    esz = 2 if dtype == PB_BF16 else 4
    d.A = A.data_ptr() + a_off * esz
    d.B = B.data_ptr() + b_off * esz
    d.C = C.data_ptr() + c_off * (4 if c_f32 else esz)
    d.bias = bias.data_ptr() if bias is not None else None
    d.aux_in = gelu_grad_aux_in.data_ptr() if gelu_grad_aux_in is not None else None
    d.aux_out = gelu_aux_out.data_ptr() if gelu_aux_out is not None else None
    d.dtype, d.a_kcontig, d.b_kcontig = dtype, int(a_kc), int(b_kc)
    d.flags = (GEMM_ACCUM if accum else 0) | (GEMM_C_F32 if c_f32 else 0) | \
              (GEMM_GELU if gelu_aux_out is not None else 0) | (GEMM_MUL_GELU_GRAD if gelu_grad_aux_in is not None else 0) | \
              (16 if force_v1 else 0) | (32 if tile128 else 0) | (64 if tile256 else 0) | dbg | _ENV_GEMM_FLAGS
    d.splitk = splitk if (splitk > 1 and slabs is not None) else 1
    d.slabs = slabs.data_ptr() if (splitk > 1 and slabs is not None) else None
    d.colsum_out = colsum_out.data_ptr() if colsum_out is not None else None
    d.colsum_ws = colsum_ws.data_ptr() if colsum_ws is not None else None
    if rowdot is not None:
        d.aux_in, d.rowdot_out, d.ld_rowdot = rowdot[0].data_ptr(), rowdot[1].data_ptr(), rowdot[2]
        d.flags |= GEMM_ROWDOT
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, nb1, nb2
    d.lda = lda if lda is not None else (K if a_kc else M)
    d.ldb = ldb if ldb is not None else (K if b_kc else N)
    d.ldc = ldc if ldc is not None else N
    d.ldaux = ldaux or d.ldc
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.alpha = alpha
    if not (A.is_cuda and B.is_cuda and C.is_cuda):
        raise PBError('gemm needs HIP device tensors; there is no CPU path')
    LIB.call('pb_gemm', ctypes.byref(d), _stream())


def linear_fwd(x, w, bias, out, dtype, **kw):
    """out (T,N) = x (T,K) @ w(N,K)^T + bias."""
    T, K = x.shape
    N = w.shape[0]
    gemm(x, w, out, M=T, N=N, K=K, dtype=dtype, bias=bias, **kw)


def ids_to_i16(ids):
    out = torch.empty(ids.shape, dtype=torch.int16, device=ids.device)
    ids = ids.contiguous()
    if ids.dtype != torch.int64:
        ids = ids.long()
    LIB.call('pb_ids_to_i16', _p(ids), _p(out), ids.numel(), _stream())
    return out


def ids_check(ids16, limits, flag):
    """flag (device int32[1]) |= 1 if any of the (..., 8) int16 ids lies outside [0, limits[column])."""
    LIB.call('pb_ids_check', _p(ids16), ids16.numel(), _p(limits), _p(flag), _stream())


def embed_ln_fwd(ids16, P, lin_bias, pos, ln_w, ln_b, y, mean, rstd, S, eps, seed, site, p_drop, padded=False, row_ids=None):
    T, d = y.shape
    if row_ids is not None:
        LIB.call('pb_embed_ln_fwd_packed', _p(ids16), _p(row_ids), _p(P), _TAB9 if padded else _SEG9, _p(lin_bias), _p(pos), _p(ln_w), _p(ln_b),
                 _p(y), _p(mean), _p(rstd), T, S, d, dtype_code(y.dtype), eps, seed, site, p_drop, _stream())
        return
    LIB.call('pb_embed_ln_fwd', _p(ids16), _p(P), _TAB9 if padded else _SEG9, _p(lin_bias), _p(pos), _p(ln_w), _p(ln_b), _p(y), _p(mean),
             _p(rstd), T, S, d, dtype_code(y.dtype), eps, seed, site, p_drop, _stream())


def embed_ln_bwd(dy, ids16, P, lin_bias, pos, ln_w, mean, rstd, dP, dpos, dbias, dgamma, dbeta, partials, S, seed, site, p_drop,
                 dz_out=None, padded=False, row_ids=None):
    T, d = dy.shape
    if row_ids is not None:
        LIB.call('pb_embed_ln_bwd_packed', _p(dy), _p(ids16), _p(row_ids), _p(P), _TAB9 if padded else _SEG9, _p(lin_bias), _p(pos), _p(ln_w),
                 _p(mean), _p(rstd), _p(dP), _p(dpos), _p(dbias), _p(dgamma), _p(dbeta), _p(partials), _p(dz_out), T, S, d,
                 dtype_code(dy.dtype), seed, site, p_drop, _stream())
        return
    LIB.call('pb_embed_ln_bwd', _p(dy), _p(ids16), _p(P), _TAB9 if padded else _SEG9, _p(lin_bias), _p(pos), _p(ln_w), _p(mean), _p(rstd),
             _p(dP), _p(dpos), _p(dbias), _p(dgamma), _p(dbeta), _p(partials), _p(dz_out), T, S, d, dtype_code(dy.dtype), seed, site,
             p_drop, _stream())


def split_bf16(x, hi, lo):
    """x (f32) -> hi = bf16(x), lo = bf16(x - hi) (pb_split_bf16)."""
    LIB.call('pb_split_bf16', _p(x), _p(hi), _p(lo), x.numel(), _stream())


def onehot_build(ids16, out, padded=False):
    LIB.call('pb_onehot_build', _p(ids16), _TAB9 if padded else _SEG9, _p(out), ids16.numel() // 8, TAB_TOTAL if padded else VOCAB, _stream())


def batch_sum(x, out, B, Sd):
    LIB.call('pb_batch_sum', _p(x), _p(out), B, Sd, dtype_code(x.dtype), _stream())


def add_ln_fwd(res, a, ln_w, ln_b, y, mean, rstd, eps, seed, site, p_drop, row_ids=None):
    T, d = y.shape
    if row_ids is not None:
        LIB.call('pb_add_ln_fwd_packed', _p(res), _p(a), _p(ln_w), _p(ln_b), _p(y), _p(mean), _p(rstd), _p(row_ids), T, d, dtype_code(y.dtype),
                 eps, seed, site, p_drop, _stream())
        return
    LIB.call('pb_add_ln_fwd', _p(res), _p(a), _p(ln_w), _p(ln_b), _p(y), _p(mean), _p(rstd), T, d, dtype_code(y.dtype),
             eps, seed, site, p_drop, _stream())


def add_ln_bwd(dy, res, a, ln_w, mean, rstd, dres, da, dgamma, dbeta, dbias_a, partials, accum_dres, seed, site, p_drop, row_ids=None):
    T, d = dy.shape
    if row_ids is not None:
        LIB.call('pb_add_ln_bwd_packed', _p(dy), _p(res), _p(a), _p(ln_w), _p(mean), _p(rstd), _p(dres), _p(da), _p(dgamma), _p(dbeta),
                 _p(dbias_a), _p(partials), _p(row_ids), T, d, dtype_code(dy.dtype), int(dres.dtype == torch.float32 and dy.dtype != torch.float32),
                 int(accum_dres), seed, site, p_drop, _stream())
        return
    LIB.call('pb_add_ln_bwd', _p(dy), _p(res), _p(a), _p(ln_w), _p(mean), _p(rstd), _p(dres), _p(da), _p(dgamma), _p(dbeta),
             _p(dbias_a), _p(partials), T, d, dtype_code(dy.dtype), int(dres.dtype == torch.float32 and dy.dtype != torch.float32),
             int(accum_dres), seed, site, p_drop, _stream())


def colsum(dy, out, partials, T, N, ld=None):
    LIB.call('pb_colsum', _p(dy), ld if ld is not None else N, _p(out), _p(partials), T, N,
             PB_F32 if dy.dtype == torch.float32 else PB_BF16, int(dy.dtype == torch.float32), _stream())


def softmax_fwd(scores, key_mask, P, B, H, Sq, Sk, scale, causal):
    LIB.call('pb_softmax_fwd', _p(scores), _p(key_mask), _p(P), B, H, Sq, Sk, scale, int(causal), dtype_code(P.dtype), _stream())


def softmax_bwd(dP, P, dS, rows, Sk, scale):
    LIB.call('pb_softmax_bwd', _p(dP), _p(P), _p(dS), rows, Sk, scale, dtype_code(P.dtype), _stream())


def ce_fwd_bwd(logits, target16, loss_mask, sums, partials, coef, dlogits, argmax_out):
    T, V = logits.shape
    LIB.call('pb_ce_fwd_bwd', _p(logits), _p(target16), _p(loss_mask), _SEG9, _p(sums), _p(partials), _p(coef), _p(dlogits),
             _p(argmax_out), T, V, dtype_code(dlogits.dtype) if dlogits is not None else PB_F32, _stream())


def mask_count(loss_mask, counts, partials):
    LIB.call('pb_mask_count', _p(loss_mask), _p(counts), _p(partials), loss_mask.numel() // 8, _stream())


def loss_coef(counts, w, coef, scale=1.0):
    LIB.call('pb_loss_coef', _p(counts), _p(w), _p(coef), scale, _stream())


def grad_sqnorm(g, partials, out_sq):
    LIB.call('pb_grad_sqnorm', _p(g), g.numel(), _p(partials), _p(out_sq), _stream())


def clip_coef(sq, max_norm, gscale, coef):
    LIB.call('pb_clip_coef', _p(sq), max_norm, gscale, _p(coef), _stream())


def adamw_step(p, g, m, v, shadow, clip, lr, beta1, beta2, eps, weight_decay, step):
    LIB.call('pb_adamw_step', _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(clip), lr, beta1, beta2, eps,
             weight_decay, step, _stream())


def cast_f32_to_bf16(src, dst):
    LIB.call('pb_cast_f32_to_bf16', _p(src), _p(dst), src.numel(), _stream())


def cast_bf16_to_f32(src, dst):
    LIB.call('pb_cast_bf16_to_f32', _p(src), _p(dst), src.numel(), _stream())


def sum_rows_bf16(src, dst, rows):
    LIB.call('pb_sum_rows_bf16', _p(src), _p(dst), rows, dst.numel(), _stream())


def transpose_batch_bf16(src, dst, table, n_tiles):
    LIB.call('pb_transpose_batch_bf16', _p(src), _p(dst), _p(table), table.shape[0], n_tiles, _stream())


def fill_f32(dst, value):
    LIB.call('pb_fill_f32', _p(dst), value, dst.numel(), _stream())


def shift_right(ids16, sos_row16, out, B, S):
    LIB.call('pb_shift_right', _p(ids16), _p(sos_row16), _p(out), B, S, _stream())


def flash_fwd(q, k, v, o, lse, key_mask, B, H, Sq, Sk, hd, scale, causal, force_generic=False, kmax=None):
    """q,k,v,o: (tensor, element offset, row stride, batch stride) bf16."""
    (qt, qo, qs, qb), (kt, ko, ks, kb), (vt, vo, vs, vb), (ot, oo, os_, ob) = q, k, v, o
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 2 * off)
    LIB.call('pb_flash_fwd', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(lse), _p(key_mask), _p(kmax), B, H, Sq, Sk, hd,
             qb, qs, kb, ks, vb, vs, ob, os_, scale, int(causal) | (2 if force_generic else 0), _stream())


def flash_bwd(q, k, v, o, dout, lse, key_mask, dq, dk, dv, delta, B, H, Sq, Sk, hd, scale, causal, force_generic=False, kmax=None, dbias=None, dbias_ws=None):
    """dbias = (gq, gk, gv): f32 vectors of H*hd that receive += the column sums of dq / dk / dv (bias gradients), with dbias_ws."""
    (qt, qo, qs, qb), (kt, ko, ks, kb), (vt, vo, vs, vb), (ot, oo, os_, ob) = q, k, v, o
    (dqt, dqo, dqs, dqb), (dkt, dko, dks, dkb), (dvt, dvo, dvs, dvb) = dq, dk, dv
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 2 * off)
    LIB.call('pb_flash_bwd', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(dout), _p(lse), _p(key_mask), _p(kmax), pp(dqt, dqo),
             pp(dkt, dko), pp(dvt, dvo), _p(delta), B, H, Sq, Sk, hd, qb, qs, kb, ks, vb, vs, ob, os_, dqb, dqs, dkb, dks, dvb, dvs,
             scale, int(causal) | (2 if force_generic else 0), _p(dbias[0]) if dbias else None, _p(dbias[1]) if dbias else None,
             _p(dbias[2]) if dbias else None, _p(dbias_ws) if dbias else None, _stream())


def flash_fwd_x3(q, k, v, o, lse, key_mask, B, H, Sq, Sk, hd, scale, causal, kmax=None):
    """Fused attention of the bf16x3 instantiation. q,k,v,o: (tensor, element offset, row stride, batch stride) f32."""
    (qt, qo, qs, qb), (kt, ko, ks, kb), (vt, vo, vs, vb), (ot, oo, os_, ob) = q, k, v, o
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)
    LIB.call('pb_flash_fwd_x3', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(lse), _p(key_mask), _p(kmax), B, H, Sq, Sk, hd,
             qb, qs, kb, ks, vb, vs, ob, os_, scale, int(causal), _stream())


def flash_bwd_x3(q, k, v, o, dout, lse, key_mask, dq, dk, dv, delta, B, H, Sq, Sk, hd, scale, causal, kmax=None):
    (qt, qo, qs, qb), (kt, ko, ks, kb), (vt, vo, vs, vb), (ot, oo, os_, ob) = q, k, v, o
    (dqt, dqo, dqs, dqb), (dkt, dko, dks, dkb), (dvt, dvo, dvs, dvb) = dq, dk, dv
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)
    LIB.call('pb_flash_bwd_x3', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(dout), _p(lse), _p(key_mask), _p(kmax), pp(dqt, dqo), pp(dkt, dko),
             pp(dvt, dvo), _p(delta), B, H, Sq, Sk, hd, qb, qs, kb, ks, vb, vs, ob, os_, dqb, dqs, dkb, dks, dvb, dvs, scale, int(causal), _stream())


def flash_fwd_x3_packed(q, k, v, o, lse, rows, B, H, hd, scale, causal):
    """q,k,v,o: (tensor, element offset, row stride) f32 over packed rows (bf16x3 instantiation)."""
    (qt, qo, qs), (kt, ko, ks), (vt, vo, vs), (ot, oo, os_) = q, k, v, o
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)
    LIB.call('pb_flash_fwd_x3_packed', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(lse), _p(rows.q_off), _p(rows.q_len), _p(rows.k_off),
             _p(rows.k_len), _p(rows.k_vis), B, H, rows.Sq_max, rows.Sk_max, hd, qs, ks, vs, os_, scale, int(causal), _stream())


def flash_bwd_x3_packed(q, k, v, o, dout, lse, dq, dk, dv, delta, rows, B, H, hd, scale, causal):
    (qt, qo, qs), (kt, ko, ks), (vt, vo, vs), (ot, oo, os_) = q, k, v, o
    (dqt, dqo, dqs), (dkt, dko, dks), (dvt, dvo, dvs) = dq, dk, dv
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)
    LIB.call('pb_flash_bwd_x3_packed', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(dout), _p(lse), pp(dqt, dqo), pp(dkt, dko), pp(dvt, dvo),
             _p(delta), _p(rows.q_off), _p(rows.q_len), _p(rows.k_off), _p(rows.k_len), _p(rows.k_vis), B, H, rows.Sq_max, rows.Sk_max, hd,
             qs, ks, vs, os_, dqs, dks, dvs, scale, int(causal), _stream())


class PackedRows:
    """Row descriptors of one packed attention call (include/pianobart_hip.h, pb_flash_*_packed): device int32 (B) tensors plus
    the two maxima."""

    def __init__(self, q_off, q_len, k_off, k_len, k_vis, Sq_max, Sk_max, kind='', order=None):
        self.q_off, self.q_len, self.k_off, self.k_len, self.k_vis, self.Sq_max, self.Sk_max = q_off, q_len, k_off, k_len, k_vis, Sq_max, Sk_max
        self.kind = kind                    # a label for profiles ('enc', 'dec', 'cross')
        self.order = order                  # device int32 (B * H): dispatch order of the (batch, head) pairs, longest first (or None)


def flash_fwd_packed(q, k, v, o, lse, rows, B, H, hd, scale, causal):
    """q,k,v,o: (tensor, element offset, row stride) bf16 over packed rows."""
    (qt, qo, qs), (kt, ko, ks), (vt, vo, vs), (ot, oo, os_) = q, k, v, o
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 2 * off)
    LIB.call('pb_flash_fwd_packed', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(lse), _p(rows.q_off), _p(rows.q_len), _p(rows.k_off),
             _p(rows.k_len), _p(rows.k_vis), B, H, rows.Sq_max, rows.Sk_max, hd, qs, ks, vs, os_, scale, int(causal), _p(rows.order), _stream())


def flash_bwd_packed(q, k, v, o, dout, lse, dq, dk, dv, delta, rows, B, H, hd, scale, causal, dbias=None, dbias_ws=None):
    (qt, qo, qs), (kt, ko, ks), (vt, vo, vs), (ot, oo, os_) = q, k, v, o
    (dqt, dqo, dqs), (dkt, dko, dks), (dvt, dvo, dvs) = dq, dk, dv
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 2 * off)
    LIB.call('pb_flash_bwd_packed', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(dout), _p(lse), pp(dqt, dqo), pp(dkt, dko), pp(dvt, dvo),
             _p(delta), _p(rows.q_off), _p(rows.q_len), _p(rows.k_off), _p(rows.k_len), _p(rows.k_vis), B, H, rows.Sq_max, rows.Sk_max, hd,
             qs, ks, vs, os_, dqs, dks, dvs, scale, int(causal), _p(dbias[0]) if dbias else None, _p(dbias[1]) if dbias else None,
             _p(dbias[2]) if dbias else None, _p(dbias_ws) if dbias else None, _p(rows.order), _stream())


_fa1_ws = {}


def _flash1_ws(nbytes, device):
    """The dQ slab workspace of the one-pass backward (one per device, grown on demand)."""
    ws = _fa1_ws.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = _fa1_ws[device] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def flash_bwd1(q, k, v, o, dout, lse, key_mask, dq, dk, dv, delta, B, H, Sq, Sk, hd, scale, causal, kmax=None, dbias=None, dbias_ws=None, delta_rows=None):
    """pb_flash_bwd1: flash_bwd's arguments and results, one pass over the (key block, query tile) pairs (head_dim 64)."""
    (qt, qo, qs, qb), (kt, ko, ks, kb), (vt, vo, vs, vb), (ot, oo, os_, ob) = q, k, v, o
    (dqt, dqo, dqs, dqb), (dkt, dko, dks, dkb), (dvt, dvo, dvs, dvb) = dq, dk, dv
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 2 * off)
    ws = _flash1_ws(int(LIB.query('pb_flash_bwd1_ws_bytes', B * Sq, H, hd, Sk)), qt.device)
    LIB.call('pb_flash_bwd1', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(dout), _p(lse), _p(key_mask), _p(kmax), pp(dqt, dqo),
             pp(dkt, dko), pp(dvt, dvo), _p(delta), B, H, Sq, Sk, hd, qb, qs, kb, ks, vb, vs, ob, os_, dqb, dqs, dkb, dks, dvb, dvs,
             scale, int(causal), _p(dbias[0]) if dbias else None, _p(dbias[1]) if dbias else None,
             _p(dbias[2]) if dbias else None, _p(dbias_ws) if dbias else None, _p(ws), _p(delta_rows), _stream())


def flash_bwd1_packed(q, k, v, o, dout, lse, dq, dk, dv, delta, rows, B, H, hd, scale, causal, q_rows, dbias=None, dbias_ws=None, delta_rows=None):
    """pb_flash_bwd1_packed: flash_bwd_packed's arguments and results in one pass; q_rows = rows of the q-side tensors."""
    (qt, qo, qs), (kt, ko, ks), (vt, vo, vs), (ot, oo, os_) = q, k, v, o
    (dqt, dqo, dqs), (dkt, dko, dks), (dvt, dvo, dvs) = dq, dk, dv
    pp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 2 * off)
    ws = _flash1_ws(int(LIB.query('pb_flash_bwd1_ws_bytes', q_rows, H, hd, rows.Sk_max)), qt.device)
    LIB.call('pb_flash_bwd1_packed', pp(qt, qo), pp(kt, ko), pp(vt, vo), pp(ot, oo), _p(dout), _p(lse), pp(dqt, dqo), pp(dkt, dko), pp(dvt, dvo),
             _p(delta), _p(rows.q_off), _p(rows.q_len), _p(rows.k_off), _p(rows.k_len), _p(rows.k_vis), B, H, rows.Sq_max, rows.Sk_max, hd,
             qs, ks, vs, os_, dqs, dks, dvs, scale, int(causal), _p(dbias[0]) if dbias else None, _p(dbias[1]) if dbias else None,
             _p(dbias[2]) if dbias else None, _p(dbias_ws) if dbias else None, _p(ws), q_rows, _p(rows.order), _p(delta_rows), _stream())


def rowmap_count(emask, dmask, loss_mask, counts):
    B, S = emask.shape
    LIB.call('pb_rowmap_count', _p(emask), _p(dmask), _p(loss_mask), _p(counts), B, S, _stream())


def rowmap_build(mask, loss_mask, off, length, row_src, row_pos, inv):
    B, S = mask.shape
    LIB.call('pb_rowmap_build', _p(mask), _p(loss_mask), _p(off), _p(length), _p(row_src), _p(row_pos), _p(inv), B, S, _stream())


def rowmap_build_sub(loss_mask, present, off, length, row_src, row_idx):
    B, S = loss_mask.shape[:2]
    LIB.call('pb_rowmap_build_sub', _p(loss_mask), _p(present), _p(off), _p(length), _p(row_src), _p(row_idx), B, S, _stream())


def scatter_rows16(src, row_dst, dst, n_rows, row_bytes):
    LIB.call('pb_scatter_rows16', _p(src), _p(row_dst), _p(dst), n_rows, row_bytes, _stream())


def gather_rows16(src, row_src, dst, n_rows, row_bytes):
    LIB.call('pb_gather_rows16', _p(src), _p(row_src), _p(dst), n_rows, row_bytes, _stream())


def pos_grad_packed(x, inv, out, B, S):
    LIB.call('pb_pos_grad_packed', _p(x), _p(inv), _p(out), B, S, x.shape[1], dtype_code(x.dtype), _stream())


def corrupt(ids16, out16, loss_mask, choice, choice_out, mask_percent, seed, pad_row, mask_row, n_tokens):
    """ids16/out16 (B,S,8) int16 device; loss_mask (B,S,8) f32; choice/choice_out (B,) int32 device or None."""
    B, S = ids16.shape[:2]
    pr = (ctypes.c_int16 * 8)(*[int(x) for x in pad_row])
    mr = (ctypes.c_int16 * 8)(*[int(x) for x in mask_row])
    nt = (ctypes.c_int32 * 8)(*[int(x) for x in n_tokens])
    LIB.call('pb_corrupt', _p(ids16), _p(out16), _p(loss_mask), _p(choice), _p(choice_out), B, S, float(mask_percent), seed, pr, mr, nt, _stream())


def corrupt_replay(ids16, out16, loss_mask, choice, mask_percent, decisions, rand_rows, pad_row, mask_row):
    """pb_corrupt with caller-supplied random decisions (layout: include/pianobart_hip.h). decisions (B, stride) int32 device,
    rand_rows (B,S,8) int16 device or None."""
    B, S = ids16.shape[:2]
    pr = (ctypes.c_int16 * 8)(*[int(x) for x in pad_row])
    mr = (ctypes.c_int16 * 8)(*[int(x) for x in mask_row])
    LIB.call('pb_corrupt_replay', _p(ids16), _p(out16), _p(loss_mask), _p(choice), B, S, float(mask_percent), _p(decisions),
             decisions.shape[1], _p(rand_rows), pr, mr, _stream())


def key_extent(key_mask, kmax):
    B, Sk = key_mask.shape
    LIB.call('pb_key_extent', _p(key_mask), _p(kmax), B, Sk, _stream())


# ---- K14: fine-tune heads (f32) ------------------------------------------------------------------------------------
def eltwise_fwd(op, x, x2, y, seed, site, p):
    LIB.call('pb_eltwise_fwd', op, _p(x), _p(x2), _p(y), x.numel(), seed, site, p, _stream())


def eltwise_bwd(op, y, x2, dy, dx, dx2, seed, site, p):
    LIB.call('pb_eltwise_bwd', op, _p(y), _p(x2), _p(dy), _p(dx), _p(dx2), dy.numel(), seed, site, p, _stream())


def softmax_dim1_fwd(x, y):
    B, S, R = x.shape
    LIB.call('pb_softmax_dim1_fwd', _p(x), _p(y), B, S, R, _stream())


def softmax_dim1_bwd(y, dy, dx):
    B, S, R = y.shape
    LIB.call('pb_softmax_dim1_bwd', _p(y), _p(dy), _p(dx), B, S, R, _stream())


def ce_rows(logits, target32, weight, coef, loss, dlogits, argmax):
    rows, C = logits.shape
    LIB.call('pb_ce_rows', _p(logits), _p(target32), _p(weight), _p(coef), _p(loss), _p(dlogits), _p(argmax), rows, C, _stream())


def gather_rows(table, ids32, bias, out):
    LIB.call('pb_gather_rows', _p(table), _p(ids32), _p(bias), _p(out), ids32.numel(), table.shape[1], table.shape[0], _stream())


def gather_rows_bwd(dout, ids32, dtable):
    LIB.call('pb_gather_rows_bwd', _p(dout), _p(ids32), _p(dtable), ids32.numel(), dtable.shape[1], dtable.shape[0], _stream())


def dropout(x, y, seed, site, p):
    LIB.call('pb_dropout', _p(x), _p(y), x.numel(), dtype_code(x.dtype), seed, site, p, _stream())


def defer_begin(arena, table):
    """Open a deferred-reduction window (K15): bias / LayerNorm-parameter partial sums collect in `arena` until defer_flush."""
    LIB.call('pb_defer_begin', _p(arena), arena.numel(), _p(table), table.numel() // int(LIB.query('pb_defer_desc_bytes')))


def defer_flush():
    LIB.call('pb_defer_flush', _stream())


def l2_penalty(p, g, weight, scratch, loss_acc):
    """finetune.py:241-243 for one parameter tensor: loss_acc += weight * ||p||, g += weight * p / ||p|| (g may be None)."""
    LIB.call('pb_l2_penalty', _p(p), _p(g), p.numel(), float(weight), _p(scratch), _p(loss_acc), _stream())
