"""Developer aid: the N = 768 GEMM shapes of the step with / without the tail split, back-to-back launches (HIP events)."""
import sys
import torch
sys.path.insert(0, '.')
from pianobart_amd import ops
from pianobart_amd._lib import PB_BF16

shapes = [(26624, 768, 3072, 'NT', 'bias'), (26624, 768, 3072, 'NN', 'accum'), (26624, 768, 2304, 'NN', 'accum'), (26624, 768, 768, 'NT', 'bias'),
          (26624, 768, 768, 'NN', 'none'), (26624, 1536, 768, 'NT', 'bias'), (26624, 2304, 768, 'NT', 'bias'),
          (32768, 768, 3072, 'NT', 'bias'), (32768, 768, 3072, 'NN', 'accum'), (32768, 768, 768, 'NT', 'bias')]
only = sys.argv[1] if len(sys.argv) > 1 else None
for M, N, K, lay, epi in shapes:
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    B = torch.randn(N, K, device='cuda').to(torch.bfloat16) if lay == 'NT' else torch.randn(K, N, device='cuda').to(torch.bfloat16)
    bias = torch.randn(N, device='cuda') if epi == 'bias' else None
    C = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    res = []
    for flags in ((0, 32768) if only is None else (int(only),)):
        for _ in range(3):
            ops.gemm(A, B, C, M=M, N=N, K=K, dtype=PB_BF16, b_kc=(lay == 'NT'), bias=bias, accum=(epi == 'accum'), dbg=flags)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, B, C, M=M, N=N, K=K, dtype=PB_BF16, b_kc=(lay == 'NT'), bias=bias, accum=(epi == 'accum'), dbg=flags)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append((ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    print('%s %6d x %5d x %5d %-6s' % (lay, M, N, K, epi), '   '.join('%7.1f us %5.0f TF' % r for r in res), flush=True)
