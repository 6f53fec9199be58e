"""How much of the one-pass backward's packed-row time is lost to partly filled 256-key blocks? The same kernel on (a) 32 sequences of 768
rows, (b) lengths in {512, 768, 1024}, (c) ragged lengths U[512, 1024] (the bench data's shape), (d) ragged lengths rounded UP to 256 --
TFLOP/s on the (query, visible key) pairs each case really has.   python tools/flash1_ragged_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
from pianobart_amd._lib import LIB
from pianobart_amd.rowpack import dispatch_order
dev, hd, B, H = 'cuda', 64, 32, 12
d = H * hd


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, qlen):
    kvis = qlen.clone()
    qoff = torch.cat([torch.zeros(1, dtype=torch.long), qlen.cumsum(0)[:-1]])
    T = int(qlen.sum())
    i32 = lambda t: t.to(torch.int32).to(dev)
    rows = ops.PackedRows(i32(qoff), i32(qlen), i32(qoff), i32(qlen), i32(kvis), int(qlen.max()), int(qlen.max()), 'enc')
    rows.order = torch.from_numpy(dispatch_order((qlen * kvis).numpy(), H)).to(dev)
    qkv = (torch.randn(T, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = (qkv, 0, 3 * d), (qkv, d, 3 * d), (qkv, 2 * d, 3 * d)
    o = torch.empty(T, d, device=dev, dtype=torch.bfloat16); do = torch.randn(T, d, device=dev).to(torch.bfloat16)
    lse = torch.empty(B, H, rows.Sq_max, device=dev); delta = torch.zeros(B, H, rows.Sq_max, device=dev)
    dqkv = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
    dq, dk, dv = (dqkv, 0, 3 * d), (dqkv, d, 3 * d), (dqkv, 2 * d, 3 * d)
    db = [torch.zeros(d, device=dev) for _ in range(3)]
    ws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, rows.Sq_max, rows.Sk_max, hd)), device=dev)
    sc = hd ** -0.5
    ops.flash_fwd_packed(q, k, v, (o, 0, d), lse, rows, B, H, hd, sc, False)
    tf = timed(lambda: ops.flash_fwd_packed(q, k, v, (o, 0, d), lse, rows, B, H, hd, sc, False))
    t1 = timed(lambda: ops.flash_bwd1_packed(q, k, v, (o, 0, d), do, lse, dq, dk, dv, delta, rows, B, H, hd, sc, False, T, dbias=db, dbias_ws=ws))
    t2 = timed(lambda: ops.flash_bwd_packed(q, k, v, (o, 0, d), do, lse, dq, dk, dv, delta, rows, B, H, hd, sc, False, dbias=db, dbias_ws=ws))
    pairs = float((qlen * kvis).sum())
    fl = 4.0 * H * hd * pairs
    blocks = int(((kvis + 255) // 256).sum()); fill = float(kvis.sum()) / (256.0 * blocks)
    print('%-34s rows %6d  key-block fill %.2f | forward %6.1f us %4.0f TF | one-pass %6.1f us %4.0f TF | pair %6.1f us %4.0f TF' % (
        name, T, fill, tf, fl / tf / 1e6, t1, 2.5 * fl / t1 / 1e6, t2, 2.5 * fl / t2 / 1e6), flush=True)


g = torch.Generator().manual_seed(1)
rag = torch.randint(512, 1025, (B,), generator=g)
case('(a) all 768', torch.full((B,), 768, dtype=torch.long))
case('(b) lengths in {512, 768, 1024}', torch.tensor([512, 768, 1024] * 11)[:B])
case('(c) ragged U[512, 1024]', rag)
case('(d) (c) rounded up to 256', (rag + 255) // 256 * 256)
case('(e) all 1024', torch.full((B,), 1024, dtype=torch.long))
