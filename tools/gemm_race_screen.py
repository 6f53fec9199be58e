"""Race screen for a change of the persistent GEMM's synchronisation structure (cdna_hip_programming.md 5: "a sync-structure edit makes a NEW template:
screen it for races over many runs at several sizes"). The K loop sums each accumulator in the same order whatever the phase schedule, so two
builds must agree BIT FOR BIT: this prints one hash per (shape, repetition) of the output of pb_gemm on fixed random operands, with other GEMMs of
the same kind in flight on a second stream (timing noise); run it under each library (PB_LIB_PATH) and diff the outputs.
  PB_LIB_PATH=$PWD/ab/head.so python tools/gemm_race_screen.py > a.txt;  python tools/gemm_race_screen.py > b.txt;  diff a.txt b.txt"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops

dev, bf = 'cuda', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(7)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
REPS = int(os.environ.get('REPS', '12'))
shapes = [(26624, 3072, 768), (26880, 768, 3072), (26624, 768, 768), (26624, 2304, 768), (26880, 768, 2304), (5120, 768, 3072), (2048, 512, 64), (2304, 768, 128),
          (4096, 4096, 4096), (26624, 1280, 768), (512, 512, 192), (33024, 768, 768)]
side = torch.cuda.Stream()
noiseA, noiseB, noiseC = rn(8192, 1024), rn(2048, 1024), torch.empty(8192, 2048, device=dev, dtype=bf)
for M, N, K in shapes:
    A, B = rn(M, K), rn(N, K)
    bias = torch.randn(N, device=dev, generator=g)
    for rep in range(REPS):
        C = torch.full((M, N), float('nan'), device=dev, dtype=bf)
        if rep % 2:                                                       # a competing GEMM on another stream for half of the repetitions
            with torch.cuda.stream(side):
                ops.gemm(noiseA, noiseB, noiseC, M=8192, N=2048, K=1024, dtype=ops.PB_BF16)
        ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, bias=bias if rep % 3 else None, dbg=(4096 if rep % 4 == 3 else 0))
        torch.cuda.synchronize()
        assert torch.isfinite(C).all()
        print(M, N, K, rep, hashlib.sha1(C.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
# the weight-gradient layout (TN, both operands row-contiguous, split-K into f32 slabs + the reduction), launched as engine._wgrad launches it
for M, N, T, nsplit in [(3072, 768, 26624, 7), (768, 3072, 26880, 7), (768, 768, 26624, 28), (2304, 768, 26624, 9), (768, 2304, 26880, 9), (512, 512, 8192, 8), (1280, 768, 26624, 17)]:
    dy, x = rn(T, M), rn(T, N)
    slabs = torch.empty(nsplit * M * N, device=dev)
    for rep in range(REPS):
        G = torch.zeros(M, N, device=dev)
        if rep % 2:
            with torch.cuda.stream(side):
                ops.gemm(noiseA, noiseB, noiseC, M=8192, N=2048, K=1024, dtype=ops.PB_BF16)
        ops.gemm(dy, x, G, M=M, N=N, K=T, dtype=ops.PB_BF16, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N, c_f32=True, splitk=nsplit, slabs=slabs, tile256=True,
                 dbg=(4096 if rep % 4 == 3 else 0))
        torch.cuda.synchronize()
        assert torch.isfinite(G).all()
        print('TN', M, N, T, rep, hashlib.sha1(G.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
# the epilogues with their own store paths: bias + GELU pair (two outputs), * gelu' (+ column sums), += and + row-dot
for M, N, K in [(26624, 3072, 768), (26880, 3072, 768), (2048, 512, 768)]:
    A, B = rn(M, K), rn(N, K)
    bias = torch.randn(N, device=dev, generator=g)
    gp = rn(M, N)
    for rep in range(max(2, REPS // 2)):
        C = torch.full((M, N), float('nan'), device=dev, dtype=bf); aux = torch.full((M, N), float('nan'), device=dev, dtype=bf)
        ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, bias=bias, gelu_aux_out=aux)
        torch.cuda.synchronize()
        assert torch.isfinite(C).all() and torch.isfinite(aux).all()
        print('GELU', M, N, K, rep, hashlib.sha1(C.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(aux.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
        C2 = torch.full((M, N), float('nan'), device=dev, dtype=bf)
        cs = torch.zeros(N, device=dev); csws = torch.empty(int(2 * ((M + 255) // 256) * N), device=dev)
        ops.gemm(A, B, C2, M=M, N=N, K=K, dtype=ops.PB_BF16, gelu_grad_aux_in=gp, colsum_out=cs, colsum_ws=csws)
        torch.cuda.synchronize()
        print('GELUGRAD', M, N, K, rep, hashlib.sha1(C2.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(cs.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
for M, N, K in [(26624, 768, 3072), (26880, 768, 2304), (26624, 768, 768)]:
    A, B = rn(M, K), rn(N, K)
    C0 = rn(M, N)
    O = rn(M, N)
    for rep in range(max(2, REPS // 2)):
        C = C0.clone()
        ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, accum=True)
        torch.cuda.synchronize()
        print('ACCUM', M, N, K, rep, hashlib.sha1(C.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
        if N == 768 and K == 768:
            C = torch.full((M, N), float('nan'), device=dev, dtype=bf); rd = torch.full((N // 64, M), float('nan'), device=dev)
            ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, rowdot=(O, rd, M))
            torch.cuda.synchronize()
            print('ROWDOT', M, N, K, rep, hashlib.sha1(C.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(rd.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)

