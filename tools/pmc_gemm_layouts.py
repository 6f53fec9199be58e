"""One launch set per layout (NT fwd, NN dgrad, TN wgrad) of the 256x256 kernels, for LDS counters:
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d out -- python3 tools/pmc_gemm_layouts.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
dev = 'cuda'
def run(M, N, K, a_kc, b_kc, sk=1, dbg=0):
    A = torch.randn((M, K) if a_kc else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev).to(torch.bfloat16)
    c32 = sk > 1
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if c32 else torch.bfloat16)
    slabs = torch.empty(sk * M * N, device=dev) if sk > 1 else None
    for _ in range(3):
        ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=c32, splitk=sk, slabs=slabs, tile256=True, dbg=dbg)
    torch.cuda.synchronize()
for dbg in (2048, 4096):
    run(4096, 4096, 4096, True, True, dbg=dbg)      # NT
    run(4096, 4096, 4096, True, False, dbg=dbg)     # NN
    run(4096, 4096, 4096, False, False, dbg=dbg)    # TN
    run(4096, 4096, 4096, False, True, dbg=dbg)     # TT
