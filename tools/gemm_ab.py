"""Same-process A/B of the 256-row ping-pong GEMM kernel with 256- and 192-wide tiles (flag bits 13 / 14), the dispatcher's own
choice and the 128x128 kernel, on the cfg-2 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops

T = int(os.environ.get('T', 32768))
dev = 'cuda'
shapes = [('NT fc1', T, 3072, 768, True, True, 1), ('NT fc1+gelu', T, 3072, 768, True, True, -1), ('NT fc2', T, 768, 3072, True, True, 1), ('NT qkv', T, 2304, 768, True, True, 1),
          ('NT out', T, 768, 768, True, True, 1), ('NT kv_c', T, 1536, 768, True, True, 1), ('NT head', T, 1280, 768, True, True, 1),
          ('NN dfc2', T, 3072, 768, True, False, 1), ('NN dfc1', T, 768, 3072, True, False, 1), ('NN dqkv', T, 768, 2304, True, False, 1),
          ('TN w1', 3072, 768, T, False, False, 7), ('TN w2', 768, 3072, T, False, False, 7), ('TN wqkv', 2304, 768, T, False, False, 9),
          ('TN wo', 768, 768, T, False, False, 28), ('NT 4k', 4096, 4096, 4096, True, True, 1), ('NT 8k', 8192, 8192, 8192, True, True, 1)]
for name, M, N, K, a_kc, b_kc, sk in shapes:
    gelu = sk < 0
    sk = abs(sk)
    A = torch.randn((M, K) if a_kc else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev).to(torch.bfloat16)
    c32 = sk > 1
    ex = dict(bias=torch.randn(N, device=dev), gelu_aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16)) if gelu else {}
    Cs = [torch.zeros(M, N, device=dev, dtype=torch.float32 if c32 else torch.bfloat16) for _ in range(4)]
    sk2 = sk if sk == 1 else max(1, 256 // (((M + 255) // 256) * ((N + 127) // 128)))
    sk3 = sk if sk == 1 else max(1, round(512 / (((M + 127) // 128) * ((N + 127) // 128))))
    slabs = torch.empty(max(sk, sk2, sk3) * M * N, device=dev) if sk > 1 else None
    fs = [lambda C=Cs[0]: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=c32, **ex, splitk=sk, slabs=slabs, tile256=True, dbg=0),
          lambda C=Cs[1]: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=c32, **ex, splitk=sk, slabs=slabs, tile256=True, dbg=65536),
          lambda C=Cs[2]: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=c32, **ex, splitk=sk, slabs=slabs, tile256=sk > 1 and M * N > 768 * 768),
          lambda C=Cs[3]: ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=c32, **ex, splitk=sk3, slabs=slabs, tile128=True)]
    for f in fs:
        f()
    torch.cuda.synchronize()
    same = torch.equal(Cs[0], Cs[1]) and (torch.equal(Cs[0], Cs[2]) or sk2 != sk)
    err = max(float((Cs[0].float() - Cs[1].float()).abs().max()), float((Cs[0].float() - Cs[2].float()).abs().max()))
    ms = [0.0, 0.0, 0.0, 0.0]
    for rep in range(5):
        for i, f in enumerate(fs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            ms[i] += e0.elapsed_time(e1) / 25
    tf = [2.0 * M * N * K / m / 1e9 for m in ms]
    print('%-8s M=%6d N=%5d K=%6d sk=%2d/%2d/%2d  256x256 %6.3f ms %5.0f TF | 256x192 %6.3f ms %5.0f TF | default %6.3f ms %5.0f TF | 128-1bar %6.3f ms %5.0f TF  same=%s maxdiff=%.3g'
          % (name, M, N, K, sk, sk2, sk3, ms[0], tf[0], ms[1], tf[1], ms[2], tf[2], ms[3], tf[3], same, err), flush=True)
