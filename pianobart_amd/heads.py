"""Autograd building blocks of the fine-tune heads (reference model.py:128-272), every one a thin wrapper over the C-ABI kernels
(pb_gemm in exact f32, pb_eltwise_*, pb_softmax_dim1_*, pb_colsum): the heads are < 0.1 % of a fine-tune step and run in f32 on top
of the backbone's hidden states, whatever precision the backbone computes in. No CPU path: tensors must live on the HIP device."""
import torch
from . import ops
from ._lib import LIB

F32 = ops.PB_F32
_seed_state = {'n': 0}


def _next_seed():
    _seed_state['n'] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_state['n'] * 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class _LinearFn(torch.autograd.Function):
    """y (R,N) = alpha * x (R,K) @ W (N,K)^T + b."""

    @staticmethod
    def forward(ctx, x, W, b, alpha):
        x2, Wc = _f32c(x).reshape(-1, x.shape[-1]), _f32c(W)
        R, K, N = x2.shape[0], x2.shape[1], Wc.shape[0]
        y = torch.empty(R, N, dtype=torch.float32, device=x2.device)
        ops.gemm(x2, Wc, y, M=R, N=N, K=K, dtype=F32, bias=_f32c(b) if b is not None else None, alpha=alpha)
        ctx.save_for_backward(x2, Wc)
        ctx.has_b, ctx.shape, ctx.alpha = b is not None, x.shape, alpha
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, Wc = ctx.saved_tensors
        R, K, N = x2.shape[0], x2.shape[1], Wc.shape[0]
        dy2 = _f32c(dy).reshape(R, N)
        dx = torch.empty(R, K, dtype=torch.float32, device=dy2.device)
        ops.gemm(dy2, Wc, dx, M=R, N=K, K=N, dtype=F32, b_kc=False, ldb=K, alpha=ctx.alpha)      # dx = alpha dy W
        dW = torch.empty(N, K, dtype=torch.float32, device=dy2.device)
        ops.gemm(dy2, x2, dW, M=N, N=K, K=R, dtype=F32, a_kc=False, b_kc=False, lda=N, ldb=K, alpha=ctx.alpha)   # dW = alpha dy^T x
        db = None
        if ctx.has_b:
            db = torch.zeros(N, dtype=torch.float32, device=dy2.device)
            partials = torch.empty(int(LIB.query('pb_colsum_partials_floats', N)), dtype=torch.float32, device=dy2.device)
            ops.colsum(dy2, db, partials, R, N)
        return dx.reshape(ctx.shape), dW, db, None


class _ActFn(torch.autograd.Function):
    """op 1 tanh, 2 relu, 3 sigmoid."""

    @staticmethod
    def forward(ctx, x, op):
        xc = _f32c(x)
        y = torch.empty_like(xc)
        ops.eltwise_fwd(op, xc, None, y, 0, 0, 0.0)
        ctx.save_for_backward(y)
        ctx.op = op
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.eltwise_bwd(ctx.op, y, None, _f32c(dy), dx, None, 0, 0, 0.0)
        return dx, None


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        xc = _f32c(x)
        y = torch.empty_like(xc)
        ops.eltwise_fwd(4, xc, None, y, seed, 0x7001, p)
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dyc = _f32c(dy)
        dx = torch.empty_like(dyc)
        ops.eltwise_bwd(4, dyc, None, dyc, dx, None, ctx.seed, 0x7001, ctx.p)
        return dx, None, None


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ac, bc = _f32c(a), _f32c(b)
        y = torch.empty_like(ac)
        ops.eltwise_fwd(5, ac, bc, y, 0, 0, 0.0)
        ctx.save_for_backward(ac, bc)
        return y

    @staticmethod
    def backward(ctx, dy):
        ac, bc = ctx.saved_tensors
        da, db = torch.empty_like(ac), torch.empty_like(ac)
        ops.eltwise_bwd(5, ac, bc, _f32c(dy), da, db, 0, 0, 0.0)
        return da, db


class _SoftmaxDim1Fn(torch.autograd.Function):
    """F.softmax(x, dim=1) for x (B, S, R)."""

    @staticmethod
    def forward(ctx, x):
        xc = _f32c(x)
        y = torch.empty_like(xc)
        ops.softmax_dim1_fwd(xc, y)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.softmax_dim1_bwd(y, _f32c(dy), dx)
        return dx


class _PoolFn(torch.autograd.Function):
    """m (B, R, d) = p (B, S, R)^T x (B, S, d): torch.bmm(attn_mat.permute(0, 2, 1)^T ..) of model.py:213."""

    @staticmethod
    def forward(ctx, p, x):
        pc, xc = _f32c(p), _f32c(x)
        B, S, R = pc.shape
        d = xc.shape[2]
        m = torch.empty(B, R, d, dtype=torch.float32, device=xc.device)
        ops.gemm(pc, xc, m, M=R, N=d, K=S, dtype=F32, a_kc=False, b_kc=False, lda=R, ldb=d, ldc=d, nb1=B, sA=(S * R, 0), sB=(S * d, 0), sC=(R * d, 0))
        ctx.save_for_backward(pc, xc)
        return m

    @staticmethod
    def backward(ctx, dm):
        pc, xc = ctx.saved_tensors
        B, S, R = pc.shape
        d = xc.shape[2]
        dmc = _f32c(dm)
        dp = torch.empty(B, S, R, dtype=torch.float32, device=xc.device)
        ops.gemm(xc, dmc, dp, M=S, N=R, K=d, dtype=F32, lda=d, ldb=d, ldc=R, nb1=B, sA=(S * d, 0), sB=(R * d, 0), sC=(S * R, 0))             # dp = x dm^T
        dx = torch.empty(B, S, d, dtype=torch.float32, device=xc.device)
        ops.gemm(pc, dmc, dx, M=S, N=d, K=R, dtype=F32, b_kc=False, lda=R, ldb=d, ldc=d, nb1=B, sA=(S * R, 0), sB=(R * d, 0), sC=(S * d, 0))  # dx = p dm
        return dp, dx


def linear(x, W, b=None, alpha=1.0):
    return _LinearFn.apply(x, W, b, alpha)


def act(x, op):
    return _ActFn.apply(x, {'tanh': 1, 'relu': 2, 'sigmoid': 3}[op])


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _DropoutFn.apply(x, p, _next_seed())


def mul(a, b):
    return _MulFn.apply(a, b)


def softmax_dim1(x):
    return _SoftmaxDim1Fn.apply(x)


def pool(p, x):
    return _PoolFn.apply(p, x)


class _CEFn(torch.autograd.Function):
    """Per-row nn.CrossEntropyLoss(reduction='none') (finetune.py:118-129); the backward re-runs the row kernel with the incoming
    per-row gradient as the row weight."""

    @staticmethod
    def forward(ctx, logits, target):
        lc = _f32c(logits).reshape(-1, logits.shape[-1])
        t32 = target.reshape(-1).to(torch.int32).contiguous()
        loss = torch.empty(lc.shape[0], dtype=torch.float32, device=lc.device)
        ops.ce_rows(lc, t32, None, None, loss, None, None)
        ctx.save_for_backward(lc, t32)
        ctx.shape = logits.shape
        return loss.reshape(logits.shape[:-1])

    @staticmethod
    def backward(ctx, dloss):
        lc, t32 = ctx.saved_tensors
        g = torch.empty_like(lc)
        tmp = torch.empty(lc.shape[0], dtype=torch.float32, device=lc.device)
        ops.ce_rows(lc, t32, _f32c(dloss).reshape(-1), None, tmp, g, None)
        return g.reshape(ctx.shape), None


def cross_entropy_rows(logits, target):
    """logits (..., C) f32, target (...) integer -> loss (...)."""
    return _CEFn.apply(logits, target)


class _GatherFn(torch.autograd.Function):
    """out (..., d) = table (n, d)[ids] + bias."""

    @staticmethod
    def forward(ctx, table, ids, bias):
        tc = _f32c(table)
        i32 = ids.reshape(-1).to(torch.int32).contiguous()
        out = torch.empty(i32.numel(), tc.shape[1], dtype=torch.float32, device=tc.device)
        ops.gather_rows(tc, i32, _f32c(bias) if bias is not None else None, out)
        ctx.save_for_backward(i32)
        ctx.n, ctx.has_b = tc.shape[0], bias is not None
        return out.reshape(*ids.shape, tc.shape[1])

    @staticmethod
    def backward(ctx, dout):
        (i32,) = ctx.saved_tensors
        dc = _f32c(dout).reshape(i32.numel(), -1)
        d = dc.shape[1]
        dt = torch.empty(ctx.n, d, dtype=torch.float32, device=dc.device)
        ops.gather_rows_bwd(dc, i32, dt)
        db = None
        if ctx.has_b:
            db = torch.zeros(d, dtype=torch.float32, device=dc.device)
            partials = torch.empty(int(LIB.query('pb_colsum_partials_floats', d)), dtype=torch.float32, device=dc.device)
            ops.colsum(dc, db, partials, dc.shape[0], d)
        return dt, None, db


def gather_rows(table, ids, bias=None):
    return _GatherFn.apply(table, ids, bias)
