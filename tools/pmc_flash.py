"""A few launches of the hd-64 attention kernels (cfg-2 shape) for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
B, H, S, hd = 32, 12, 1024, 64
d = H * hd
dev = 'cuda'
qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(B * S, d, device=dev, dtype=torch.bfloat16); do = torch.randn(B * S, d, device=dev).to(torch.bfloat16)
dqkv = torch.empty(B * S, 3 * d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev); delta = torch.empty(B, H, S, device=dev)
q = (qkv, 0, 3 * d, S * 3 * d); k = (qkv, d, 3 * d, S * 3 * d); v = (qkv, 2 * d, 3 * d, S * 3 * d); oo = (o, 0, d, S * d)
dq = (dqkv, 0, 3 * d, S * 3 * d); dk = (dqkv, d, 3 * d, S * 3 * d); dv = (dqkv, 2 * d, 3 * d, S * 3 * d)
for _ in range(3):
    ops.flash_fwd(q, k, v, oo, lse, None, B, H, S, S, hd, hd ** -0.5, False)
    ops.flash_bwd(q, k, v, oo, do, lse, None, dq, dk, dv, delta, B, H, S, S, hd, hd ** -0.5, False)
    ops.flash_bwd1(q, k, v, oo, do, lse, None, dq, dk, dv, delta, B, H, S, S, hd, hd ** -0.5, False)
torch.cuda.synchronize()
