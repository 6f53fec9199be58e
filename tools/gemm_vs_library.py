"""Known-good reference for the GEMM shapes of the step (cdna_hip_programming.md 5.4 rule 10: a ceiling claim needs a reference measured on the
same hardware): torch.matmul (hipBLASLt / rocBLAS behind it) against pb_gemm on the same random bf16 operands, plain epilogues only (the library
has no GELU-pair / gelu' / row-dot epilogue). MEASUREMENT ONLY: nothing in the product calls a library GEMM.
  python tools/gemm_vs_library.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 26624
dev, bf = 'cuda', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(2)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


print('%-28s %10s %10s   %s' % ('shape (rows = %d)' % T, 'pb_gemm us', 'library us', 'TFLOP/s ours / library'))
for name, M, N, K, lay in (('NT fc1 (no epilogue)', T, 3072, 768, 'NT'), ('NT qkv', T, 2304, 768, 'NT'), ('NT out-proj', T, 768, 768, 'NT'), ('NT fc2', T, 768, 3072, 'NT'),
                           ('NT dqkv', T, 768, 2304, 'NT'), ('TN w1', 3072, 768, T, 'TN'), ('TN wqkv', 2304, 768, T, 'TN'), ('TN wo', 768, 768, T, 'TN'),
                           ('NT 4096^3', 4096, 4096, 4096, 'NT'), ('NT 8192^3', 8192, 8192, 8192, 'NT')):
    fl = 2.0 * M * N * K
    if lay == 'NT':
        A, B = rn(M, K), rn(N, K)
        C = torch.empty(M, N, device=dev, dtype=bf)
        ours = timeit(lambda: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16))
        Bt = B.t()
        lib = timeit(lambda: torch.matmul(A, Bt, out=C))
    else:                                                                  # C (M, N) = A^T B with A (K, M), B (K, N): the weight-gradient layout, f32 result
        A, B = rn(K, M), rn(K, N)
        C = torch.empty(M, N, device=dev, dtype=torch.float32)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        sk = max(1, min(32, round(192 / tiles)))
        slabs = torch.empty(sk * M * N, device=dev)
        ours = timeit(lambda: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=False, b_kc=False, c_f32=True, splitk=sk, slabs=slabs, tile256=True))
        At = A.t()
        Cb = torch.empty(M, N, device=dev, dtype=bf)
        lib = timeit(lambda: torch.matmul(At, B, out=Cb))                  # bf16 result: less to write than ours
    print('%-28s %10.1f %10.1f   %6.0f / %6.0f' % (name, ours, lib, fl / ours / 1e6, fl / lib / 1e6), flush=True)
