"""Per-phase cycle anatomy of the one-pass backward's 32-query step (diagnostic build of the library:
PB_EXTRA_HIPCC_FLAGS=-DPB_FA1_STAMPS python pianobart_amd/build.py --force, into a separate PB_LIB_PATH). Means over every wave and step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(32, dtype=torch.int32, device='cuda')
os.environ['PB_FA1_STAMP_PTR'] = str(st.data_ptr())
from tools import flash1_check as F
kind = sys.argv[1] if len(sys.argv) > 1 else 'dense'
# timed loops (back-to-back launches, clocks up): every figure below is a mean per wave or per step over all of them
(F.dense(32, 12, 1024, None, kind == 'causal', True) if kind in ('dense', 'causal') else F.packed(32, 12, 1024, kind, True, ordered=True))
torch.cuda.synchronize()
v = st.cpu().numpy().astype('uint32').astype('float64')
names = ['barrier exit -> requests', 'requests + dV/dK(prev) 16 MFMA', 'wait A', 'S/dP A 16 MFMA (+ B requests)', 'wait B + transposed', 'S/dP B 16 MFMA || softmax A',
         'dV/dK 16 + dQ 16 MFMA || softmax B', 'hand-over + dQ stores', 'waits in front of barrier', 'barrier']
n = v[16]
v[:13] *= 16; v[18:32] *= 16                                     # the kernel adds cycles / 16 (32-bit sums)
print('steps x waves:', int(n), ' waves:', int(v[17]))
tot = 0
for i, nm in enumerate(names):
    print('%-40s %8.0f cycles / step' % (nm, v[i] / n)); tot += v[i] / n
print('%-40s %8.0f   (0 everywhere: coarse build, -DPB_FA1_STAMPS=1)' % ('sum', tot))
w = v[17]
print('per workgroup-wave: prologue %.0f cycles, steps %.0f (%.0f per step), epilogue %.0f' % (v[10] / w, v[11] / w, v[11] / n, v[12] / w))
print('prologue: loads + DMA issue %.0f | tables %.0f | K/V fragments + accumulator init %.0f | wait vmcnt(0) %.0f | barrier %.0f | K^T fragments %.0f | rest + barrier %.0f' % tuple(v[18:25] / w))
print('loads + DMA issue: K / V fragment loads %.0f | -lse / -delta loads %.0f | descriptors, K-image DMA %.0f | first three {Q, dO} tiles %.0f' % tuple(v[28:32] / w))
print('epilogue: last step %.0f | dK / dV rows %.0f | bias partials %.0f' % tuple(v[25:28] / w))
