"""Runs ONLY the dominant kernel (bf16 NT GEMM at the fc1 shape of cfg 2, with the epilogue the step launches it with: bias + GELU +
derivative out) a few times, for rocprofv3 --pmc passes:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_gemm.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_gemm.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
T, N, K = 32768, 3072, 768
for a in sys.argv[1:]:
    if a.startswith('--M='):
        T = int(a[4:])                      # packed step: the encoder / decoder side row counts of the bench batch
x = torch.randn(T, K, device='cuda').to(torch.bfloat16); w = torch.randn(N, K, device='cuda').to(torch.bfloat16)
out = torch.empty(T, N, device='cuda', dtype=torch.bfloat16); aux = torch.empty_like(out)
bias = torch.randn(N, device='cuda')
plain = '--plain' in sys.argv
for _ in range(6):
    if plain:
        ops.gemm(x, w, out, M=T, N=N, K=K, dtype=ops.PB_BF16)
    else:
        ops.gemm(x, w, out, M=T, N=N, K=K, dtype=ops.PB_BF16, bias=bias, gelu_aux_out=aux)
torch.cuda.synchronize()
