"""Times the head_dim-64 attention kernels at the cfg-2 shape (B=32, H=12, S=1024) with rocprof-free HIP events:
forward / backward, unmasked / ragged key mask (+ key extent) / causal. FLOPs counted for the visible part only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops

B, H, S, hd = 32, 12, 1024, 64
d = H * hd
dev = 'cuda'
torch.manual_seed(0)
qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(B * S, d, device=dev, dtype=torch.bfloat16)
do = torch.randn(B * S, d, device=dev).to(torch.bfloat16)
dqkv = torch.empty(B * S, 3 * d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev); delta = torch.empty(B, H, S, device=dev)
lens = torch.randint(S // 2, S + 1, (B,), device=dev)
mask = (torch.arange(S, device=dev)[None, :] < lens[:, None]).float().contiguous()
kmax = torch.empty(B, dtype=torch.int32, device=dev); ops.key_extent(mask, kmax)
q = (qkv, 0, 3 * d, S * 3 * d); k = (qkv, d, 3 * d, S * 3 * d); v = (qkv, 2 * d, 3 * d, S * 3 * d); oo = (o, 0, d, S * d)
dq = (dqkv, 0, 3 * d, S * 3 * d); dk = (dqkv, d, 3 * d, S * 3 * d); dv = (dqkv, 2 * d, 3 * d, S * 3 * d)
scale = hd ** -0.5


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


full = 4.0 * S * S * hd * B * H
vis = float((lens.double() / S).mean())             # visible key fraction under the ragged mask
for name, km, kx, causal, frac in [('unmasked', None, None, False, 1.0), ('ragged mask+extent', mask, kmax, False, vis), ('causal', None, None, True, 0.5 + 0.5 * 64 / S),
                                   ('causal+ragged', mask, kmax, True, 0.5)]:
    tf = timed(lambda: ops.flash_fwd(q, k, v, oo, lse, km, B, H, S, S, hd, scale, causal, kmax=kx))
    tb = timed(lambda: ops.flash_bwd(q, k, v, oo, do, lse, km, dq, dk, dv, delta, B, H, S, S, hd, scale, causal, kmax=kx))
    print('%-20s fwd %7.1f us (%5.0f TF on the visible part) | bwd %7.1f us (%5.0f TF)' % (name, tf * 1e3, full * frac / tf / 1e9, tb * 1e3, 2.5 * full * frac / tb / 1e9), flush=True)
