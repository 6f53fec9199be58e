"""Times of the fused split-bf16 attention (pb_flash_x3.hip) at the cfg-2 shape of the bf16x3 parity step (B = 16, H = 12, S = 1024, head_dim 64, f32): forward and
backward (delta + dK/dV + dQ), unmasked and causal. HIP events, 10 launches.  python tools/flash_x3_bench.py   (PB_LIB_PATH selects another build for an A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops

B, H, S, hd = 16, 12, 1024, 64
d = H * hd
dev = 'cuda'
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * d, device=dev) * 0.5
o = torch.empty(B * S, d, device=dev); do = torch.randn(B * S, d, device=dev)
dqkv = torch.empty(B * S, 3 * d, device=dev)
lse = torch.empty(B, H, S, device=dev); delta = torch.empty(B, H, S, device=dev)
q = (qkv, 0, 3 * d, S * 3 * d); k = (qkv, d, 3 * d, S * 3 * d); v = (qkv, 2 * d, 3 * d, S * 3 * d); oo = (o, 0, d, S * d)
dq = (dqkv, 0, 3 * d, S * 3 * d); dk = (dqkv, d, 3 * d, S * 3 * d); dv = (dqkv, 2 * d, 3 * d, S * 3 * d)


def timed(f, n=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for causal in (False, True):
    tf = timed(lambda: ops.flash_fwd_x3(q, k, v, oo, lse, None, B, H, S, S, hd, hd ** -0.5, causal))
    tb = timed(lambda: ops.flash_bwd_x3(q, k, v, oo, do, lse, None, dq, dk, dv, delta, B, H, S, S, hd, hd ** -0.5, causal))
    print('causal=%d: forward %.1f us, backward (delta + dK/dV + dQ) %.1f us' % (causal, tf, tb), flush=True)
