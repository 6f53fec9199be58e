"""Microbenchmark of the GEMM shapes of the cfg-2 pre-train step (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops

T = 32768
dev = 'cuda'
shapes = [('NT fc1', T, 3072, 768, True, True), ('NT fc2', T, 768, 3072, True, True), ('NT qkv', T, 2304, 768, True, True),
          ('NT out', T, 768, 768, True, True), ('NT head', T, 1280, 768, True, True),
          ('NN dfc2', T, 3072, 768, True, False), ('NN dfc1', T, 768, 3072, True, False), ('NN dqkv', T, 768, 2304, True, False),
          ('TN w1', 3072, 768, T, False, False), ('TN w2', 768, 3072, T, False, False), ('TN wqkv', 2304, 768, T, False, False),
          ('TN wo', 768, 768, T, False, False)]


def run(name, M, N, K, a_kc, b_kc, force_v1, splitk=1, tile128=False):
    tile256 = not tile128
    A = torch.randn((M, K) if a_kc else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev).to(torch.bfloat16)
    c32 = not (a_kc)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if c32 else torch.bfloat16)
    slabs = torch.empty(splitk * M * N, device=dev) if splitk > 1 else None
    f = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=c32, force_v1=force_v1, splitk=splitk, slabs=slabs, tile256=tile256)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms, 2.0 * M * N * K / ms / 1e9


def run2(name, M, N, K, a_kc, b_kc, **kw):
    A = torch.randn((M, K) if a_kc else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, **kw)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return ms, 2.0 * M * N * K / ms / 1e9


for name, M, N, K, a_kc, b_kc in shapes:
    if not a_kc:
        continue
    r = [run2(name, M, N, K, a_kc, b_kc, tile128=True), run2(name, M, N, K, a_kc, b_kc), run2(name, M, N, K, a_kc, b_kc, tile256=True)]
    print('%-8s M=%6d N=%5d K=%6d  128x128 %6.3f ms %5.0f TF | 256x128 %6.3f ms %5.0f TF | 256x256 %6.3f ms %5.0f TF' % (name, M, N, K, r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1]), flush=True)
