"""Where does a K=768 GEMM launch spend its time? (profiling flags: 128 = no stores, 256 = no main loop)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
T = 32768
def run(M, N, K, dbg, t256=False, n=20):
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16); B = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, tile256=t256, dbg=dbg)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N) in ((T, 768), (T, 3072)):
    for K in (64, 768, 1536, 3072):
        for t128 in (True, False):
            full, nost, nolp = run(M, N, K, 0, t128), run(M, N, K, 128, t128), run(M, N, K, 256, t128)
            print('M=%d N=%d K=%d tile%s: full %.1f us (%.0f TF) | no-store %.1f | epilogue-only %.1f' % (M, N, K, 256 if t128 else 128, full, 2.0*M*N*K/full/1e6, nost, nolp), flush=True)

