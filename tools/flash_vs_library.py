"""Known-good reference for the head_dim-64 attention kernels (cdna_hip_programming.md 5.4 rule 10): torch.nn.functional.scaled_dot_product_attention
(the flash backend shipped with this PyTorch-ROCm build) against pb_flash_fwd / pb_flash_bwd (kernel pair) / pb_flash_bwd1 (one pass) at the cfg-2 shape
(B = 32, H = 12, S = 1024, bf16), dense, unmasked and causal. MEASUREMENT ONLY: nothing in the product calls the library.
  python tools/flash_vs_library.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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
q = (qkv, 0, 3 * d, S * 3 * d); k = (qkv, d, 3 * d, S * 3 * d); v = (qkv, 2 * d, 3 * d, S * 3 * d); oo = (o, 0, d, S * d)
dq = (dqkv, 0, 3 * d, S * 3 * d); dk = (dqkv, d, 3 * d, S * 3 * d); dv = (dqkv, 2 * d, 3 * d, S * 3 * d)
scale = hd ** -0.5


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


full = 4.0 * S * S * hd * B * H
x = qkv.view(B, S, 3, H, hd)
tq, tk, tv = (x[:, :, i].permute(0, 2, 1, 3).contiguous().requires_grad_(True) for i in range(3))       # (B, H, S, hd)
tdo = do.view(B, S, H, hd).permute(0, 2, 1, 3).contiguous()
print('%-10s %28s %28s' % ('', 'forward us (TFLOP/s)', 'backward us (TFLOP/s)'))
for name, causal, frac in (('unmasked', False, 1.0), ('causal', True, 0.5 + 0.5 * 64 / S)):
    f_ours = timed(lambda: ops.flash_fwd(q, k, v, oo, lse, None, B, H, S, S, hd, scale, causal))
    b_pair = timed(lambda: ops.flash_bwd(q, k, v, oo, do, lse, None, dq, dk, dv, delta, B, H, S, S, hd, scale, causal))
    b_one = timed(lambda: ops.flash_bwd1(q, k, v, oo, do, lse, None, dq, dk, dv, delta, B, H, S, S, hd, scale, causal))
    f_lib = timed(lambda: F.scaled_dot_product_attention(tq, tk, tv, is_causal=causal))
    out = F.scaled_dot_product_attention(tq, tk, tv, is_causal=causal)

    def lib_bwd():
        tq.grad = tk.grad = tv.grad = None
        out.backward(tdo, retain_graph=True)
    b_lib = timed(lib_bwd)
    tf = lambda us, m=1.0: m * full * frac / us / 1e6
    print('%-10s ours %6.1f (%4.0f)  library %6.1f (%4.0f) | pair %6.1f (%4.0f)  one-pass %6.1f (%4.0f)  library %6.1f (%4.0f)' %
          (name, f_ours, tf(f_ours), f_lib, tf(f_lib), b_pair, tf(b_pair, 2.5), b_one, tf(b_one, 2.5), b_lib, tf(b_lib, 2.5)), flush=True)
