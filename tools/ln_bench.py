"""Streaming-kernel rates at the cfg-2 shape (T = 32768 rows of d = 768 bf16): LayerNorm fwd / bwd with and without dropout, colsum,
against a plain device copy of the same bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
from pianobart_amd._lib import LIB
T, d = 32768, 768
dev = 'cuda'
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
NSET = 8           # rotate over 8 buffer sets (1.2-2.4 GB) so that nothing is served by the 256 MB Infinity Cache
sets = [[bf(T, d) for _ in range(6)] for _ in range(NSET)]
res, a, y, dy, dres, da = sets[0]
cnt = {'i': 0}
def nxt():
    cnt['i'] = (cnt['i'] + 1) % NSET
    return sets[cnt['i']]
w, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
gw, gb, gbias = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)
partials = torch.empty(int(LIB.query('pb_ln_partials_floats', d)), device=dev)
big = bf(T, 3072); cs = torch.zeros(3072, device=dev); pc = torch.empty(int(LIB.query('pb_colsum_partials_floats', 3072)), device=dev)


def timed(f, n=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


MB = T * d * 2 / 1e6
for p in (0.0, 0.1):
    def f_fwd():
        r_, a_, y_, _, _, _ = nxt()
        ops.add_ln_fwd(r_, a_, w, b, y_, mean, rstd, 1e-5, 1234, 7, p)
    t = timed(f_fwd)
    print('add_ln_fwd p=%.1f  %6.1f us  %.2f TB/s (3 x %.0f MB)' % (p, t, 3 * MB / t, MB))
    def f_bwd():
        r_, a_, _, dy_, dres_, da_ = nxt()
        ops.add_ln_bwd(dy_, r_, a_, w, mean, rstd, dres_, da_ if p > 0 else None, gw, gb, gbias, partials, False, 1234, 7, p)
    t = timed(f_bwd)
    n = 5 if p > 0 else 4
    print('add_ln_bwd p=%.1f  %6.1f us  %.2f TB/s (%d x %.0f MB)' % (p, t, n * MB / t, n, MB))
def f_copy():
    r_, _, y_, _, _, _ = nxt()
    y_.copy_(r_)
t = timed(f_copy)
print('device copy        %6.1f us  %.2f TB/s (2 x %.0f MB)' % (t, 2 * MB / t, MB))
t = timed(lambda: ops.colsum(big, cs, pc, T, 3072))
print('colsum T x 3072    %6.1f us  %.2f TB/s' % (t, T * 3072 * 2 / 1e6 / t))
