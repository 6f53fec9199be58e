"""One-pass attention backward (pb_flash_bwd1*) against the two-kernel backward on the same inputs: dense (unmasked, ragged key
mask, causal) and packed rows (self, causal, cross). Prints max abs differences and times (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
from pianobart_amd._lib import LIB

dev = 'cuda'
torch.manual_seed(0)
hd = 64


def timed(f, n=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def cmp(name, a, b):
    a, b = a.float(), b.float()
    d = (a - b).abs().max().item()
    print('   %-6s max|diff| %.3e  (max|ref| %.3e)' % (name, d, b.abs().max().item()), 'nan!' if not torch.isfinite(a).all() else '')
    return d


def dense(B, H, S, mask_kind, causal, time_it=True):
    d = H * hd
    qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
    o = torch.empty(B * S, d, device=dev, dtype=torch.bfloat16)
    do = torch.randn(B * S, d, device=dev).to(torch.bfloat16)
    lse = torch.empty(B, H, S, device=dev); delta = torch.empty(B, H, S, device=dev); delta1 = torch.empty(B, H, S, device=dev)
    km = kx = None
    if mask_kind == 'ragged':
        lens = torch.randint(S // 2, S + 1, (B,), device=dev)
        km = (torch.arange(S, device=dev)[None, :] < lens[:, None]).float().contiguous()
    elif mask_kind == 'scattered':
        km = (torch.rand(B, S, device=dev) > 0.3).float().contiguous()
        km[0] = 0                                              # a sample without a visible key
    if km is not None:
        kx = torch.empty(B, dtype=torch.int32, device=dev); ops.key_extent(km, kx)
    q = (qkv, 0, 3 * d, S * 3 * d); k = (qkv, d, 3 * d, S * 3 * d); v = (qkv, 2 * d, 3 * d, S * 3 * d); oo = (o, 0, d, S * d)
    scale = hd ** -0.5
    ops.flash_fwd(q, k, v, oo, lse, km, B, H, S, S, hd, scale, causal, kmax=kx)
    res = []
    for fn, dl in ((ops.flash_bwd, delta), (ops.flash_bwd1, delta1)):
        dqkv = torch.full((B * S, 3 * d), float('nan'), device=dev, dtype=torch.bfloat16)
        dq = (dqkv, 0, 3 * d, S * 3 * d); dk = (dqkv, d, 3 * d, S * 3 * d); dv = (dqkv, 2 * d, 3 * d, S * 3 * d)
        db = [torch.zeros(d, device=dev) for _ in range(3)]
        ws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, S, S, hd)), device=dev)
        run = lambda: fn(q, k, v, oo, do, lse, km, dq, dk, dv, dl, B, H, S, S, hd, scale, causal, kmax=kx, dbias=db, dbias_ws=ws)
        run(); torch.cuda.synchronize()
        db = [x.clone() for x in db]
        t = timed(run) if time_it else 0.0
        res.append((dqkv, db, t))
    (r0, b0, t0), (r1, b1, t1) = res
    print('dense B=%d H=%d S=%d mask=%s causal=%d: two-kernel %.1f us, one-pass %.1f us' % (B, H, S, mask_kind, causal, t0, t1))
    cmp('delta', delta1, delta)
    for i, n in enumerate(('dq', 'dk', 'dv')):
        cmp(n, r1[:, i * d:(i + 1) * d], r0[:, i * d:(i + 1) * d])
    for i, n in enumerate(('dbq', 'dbk', 'dbv')):
        cmp(n, b1[i], b0[i])


def packed(B, H, S, kind, time_it=True, ordered=False):
    d = H * hd
    g = torch.Generator().manual_seed(1)
    qlen = torch.randint(S // 2, S + 1, (B,), generator=g)
    klen = qlen.clone() if kind != 'cross' else torch.randint(S // 2, S + 1, (B,), generator=g)
    kvis = (klen - torch.randint(0, 40, (B,), generator=g)).clamp(min=1)
    if kind == 'dec':
        kvis = klen - torch.randint(0, 20, (B,), generator=g)           # causal decoder: visible prefix, a few loss-only rows behind it
    qoff = torch.cat([torch.zeros(1, dtype=torch.long), qlen.cumsum(0)[:-1]]); koff = torch.cat([torch.zeros(1, dtype=torch.long), klen.cumsum(0)[:-1]])
    Tq, Tk = int(qlen.sum()), int(klen.sum())
    i32 = lambda t: t.to(torch.int32).to(dev)
    rows = ops.PackedRows(i32(qoff), i32(qlen), i32(koff), i32(klen), i32(kvis), int(qlen.max()), int(klen.max()), kind)
    causal = kind == 'dec'
    if ordered:
        from pianobart_amd.rowpack import dispatch_order
        cost = (kvis * kvis // 2 + (qlen - kvis).clamp(min=0) * kvis) if causal else qlen * kvis
        rows.order = torch.from_numpy(dispatch_order(cost.numpy(), H)).to(dev)
    scale = hd ** -0.5
    if kind == 'cross':
        qb = (torch.randn(Tq, d, device=dev) * 0.5).to(torch.bfloat16); kvb = (torch.randn(Tk, 2 * d, device=dev) * 0.5).to(torch.bfloat16)
        q = (qb, 0, d); k = (kvb, 0, 2 * d); v = (kvb, d, 2 * d)
    else:
        qkv = (torch.randn(Tq, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
        q = (qkv, 0, 3 * d); k = (qkv, d, 3 * d); v = (qkv, 2 * d, 3 * d)
    o = torch.empty(Tq, d, device=dev, dtype=torch.bfloat16); do = torch.randn(Tq, d, device=dev).to(torch.bfloat16)
    lse = torch.empty(B, H, rows.Sq_max, device=dev); delta = torch.zeros(B, H, rows.Sq_max, device=dev); delta1 = torch.zeros(B, H, rows.Sq_max, device=dev)
    ops.flash_fwd_packed(q, k, v, (o, 0, d), lse, rows, B, H, hd, scale, causal)
    tfwd = timed(lambda: ops.flash_fwd_packed(q, k, v, (o, 0, d), lse, rows, B, H, hd, scale, causal)) if time_it else 0.0
    res = []
    for one, dl in ((False, delta), (True, delta1)):
        if kind == 'cross':
            dqb = torch.full((Tq, d), float('nan'), device=dev, dtype=torch.bfloat16); dkvb = torch.full((Tk, 2 * d), float('nan'), device=dev, dtype=torch.bfloat16)
            dq = (dqb, 0, d); dk = (dkvb, 0, 2 * d); dv = (dkvb, d, 2 * d)
            outs = lambda: (dqb, dkvb[:, :d], dkvb[:, d:])
        else:
            dqkv = torch.full((Tq, 3 * d), float('nan'), device=dev, dtype=torch.bfloat16)
            dq = (dqkv, 0, 3 * d); dk = (dqkv, d, 3 * d); dv = (dqkv, 2 * d, 3 * d)
            outs = lambda: (dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:])
        db = [torch.zeros(d, device=dev) for _ in range(3)]
        ws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, rows.Sq_max, rows.Sk_max, hd)), device=dev)
        if one:
            run = lambda: ops.flash_bwd1_packed(q, k, v, (o, 0, d), do, lse, dq, dk, dv, dl, rows, B, H, hd, scale, causal, Tq, dbias=db, dbias_ws=ws)
        else:
            run = lambda: ops.flash_bwd_packed(q, k, v, (o, 0, d), do, lse, dq, dk, dv, dl, rows, B, H, hd, scale, causal, dbias=db, dbias_ws=ws)
        run(); torch.cuda.synchronize()
        db = [x.clone() for x in db]
        t = timed(run) if time_it else 0.0
        res.append((outs(), db, t))
    (r0, b0, t0), (r1, b1, t1) = res
    print('packed %s B=%d H=%d S=%d (Tq %d, Tk %d)%s: forward %.1f us, backward two-kernel %.1f us, one-pass %.1f us' % (
        kind, B, H, S, Tq, Tk, ' longest first' if ordered else '', tfwd, t0, t1))
    for i, n in enumerate(('dq', 'dk', 'dv')):
        cmp(n, r1[i], r0[i])
    for i, n in enumerate(('dbq', 'dbk', 'dbv')):
        cmp(n, b1[i], b0[i])


if __name__ == '__main__':
    small = '--small' in sys.argv
    if small:
        dense(2, 2, 200, None, False, False); dense(2, 2, 333, 'ragged', False, False); dense(3, 2, 520, 'scattered', False, False); dense(2, 2, 300, None, True, False)
        packed(3, 2, 400, 'enc', False); packed(3, 2, 400, 'dec', False); packed(3, 2, 400, 'cross', False); packed(4, 2, 400, 'enc', False, ordered=True); packed(4, 2, 400, 'dec', False, ordered=True)
    else:
        dense(32, 12, 1024, None, False); dense(32, 12, 1024, 'ragged', False); dense(32, 12, 1024, None, True); dense(32, 12, 1024, 'ragged', True)
        for kind in ('enc', 'dec', 'cross'):
            packed(32, 12, 1024, kind); packed(32, 12, 1024, kind, ordered=True)
