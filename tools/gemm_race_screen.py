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
