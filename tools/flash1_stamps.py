"""Anatomy of the one-pass attention backward from the diagnostic build of the library
(PB_LIB_OUT=ab/stamps.so PB_EXTRA_HIPCC_FLAGS=-DPB_FA1_STAMPS python pianobart_amd/build.py; -DPB_FA1_STAMPS=2 adds the per-phase stamps of the 32-query step):
  PB_LIB_PATH=ab/stamps.so python tools/flash1_stamps.py dense|causal|enc|dec|cross            per-phase / prologue / epilogue cycle means over every wave (the sums are atomics: the kernel runs ~4x slower)
  PB_LIB_PATH=ab/stamps.so python tools/flash1_stamps.py dense|causal|enc|dec|cross --gaps     WORKGROUP TRACE: CU (HW_ID / XCC_ID), entry and exit time of every workgroup of the last launch --
                                                                                                idle time per CU between two workgroups, workgroup -> shader engine table (profiles/r06_attention_dispatch_trace.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
GAPS = '--gaps' in sys.argv                                     # workgroup trace instead of the phase sums: per CU, the idle time between one workgroup's exit and the next one's entry
if GAPS:
    sys.argv.remove('--gaps')
NREC = 4096
st = torch.zeros(64 + 8 * NREC if GAPS else 32, dtype=torch.int32, device='cuda')
if GAPS:
    st[32] = NREC
os.environ['PB_FA1_STAMP_PTR'] = str(st.data_ptr())
from tools import flash1_check as F
kind = sys.argv[1] if len(sys.argv) > 1 else 'dense'
# timed loops (back-to-back launches, clocks up): every figure below is a mean per wave or per step over all of them
(F.dense(32, 12, 1024, None, kind == 'causal', True) if kind in ('dense', 'causal') else F.packed(32, 12, 1024, kind, True, ordered=True))
torch.cuda.synchronize()
if GAPS:
    import numpy as np
    rec = st[64:].cpu().numpy().view(np.uint64).reshape(NREC, 4)
    rec = rec[rec[:, 1] != 0]                                    # the last launch's records (every launch overwrites them)
    cu = (rec[:, 0] >> np.uint64(32)) * np.uint64(1 << 16) + ((rec[:, 0] >> np.uint64(8)) & np.uint64(0xf)) + ((rec[:, 0] >> np.uint64(12)) & np.uint64(1)) * np.uint64(16) + ((rec[:, 0] >> np.uint64(13)) & np.uint64(7)) * np.uint64(32)
    t_in, t_out = rec[:, 1].astype(np.int64) * 24, rec[:, 2].astype(np.int64) * 24          # s_memrealtime: 10 ns ticks -> cycles at 2.4 GHz
    pro, epi = (rec[:, 3] >> np.uint64(32)).astype(np.int64), (rec[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    t0 = t_in.min()
    span = t_out.max() - t0
    gaps, busy, first, last = [], [], [], []
    for c in np.unique(cu):
        m = cu == c
        o = np.argsort(t_in[m]); a, b = t_in[m][o], t_out[m][o]
        gaps += list(a[1:] - b[:-1]); busy.append((b - a).sum()); first.append(a[0] - t0); last.append(t_out.max() - b[-1])
    gaps = np.array(gaps)
    xc = (rec[:, 0] >> np.uint64(32)).astype(np.int64)
    util, tails = [], []
    for x_ in np.unique(xc):                                      # the counters of different XCDs are not synchronised: spans per XCD
        m = xc == x_
        sp = t_out[m].max() - t_in[m].min()
        ncu = len(np.unique(cu[m]))
        util.append((t_out[m] - t_in[m]).sum() / (ncu * sp))
        tails.append(np.mean([t_out[m].max() - t_out[m][cu[m] == c].max() for c in np.unique(cu[m])]) / sp)
    print('%d workgroups on %d CUs; per XCD: CUs hold a workgroup %.1f %% of the kernel span (min %.1f, max %.1f), a CU idles %.1f %% of it behind its last workgroup' % (
        len(rec), len(np.unique(cu)), 100 * np.mean(util), 100 * min(util), 100 * max(util), 100 * np.mean(tails)))
    print('per workgroup: lifetime mean %.0f | prologue %.0f | epilogue %.0f' % ((t_out - t_in).mean(), pro.mean(), epi.mean()))
    print('per CU: gap exit -> next entry: mean %.0f, median %.0f, p90 %.0f (x %.2f per CU)' % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), len(gaps) / len(np.unique(cu))))
    idx = np.nonzero(st[64:].cpu().numpy().view(np.uint64).reshape(NREC, 4)[:, 1] != 0)[0]          # blockIdx of each record
    se = ((rec[:, 0] >> np.uint64(13)) & np.uint64(7)).astype(np.int64); xcc = (rec[:, 0] >> np.uint64(32)).astype(np.int64)
    slot = idx >> 3
    print('XCC == blockIdx & 7 for %.1f %% of the workgroups; shader engine ids seen: %s' % (100 * np.mean(xcc == (idx & 7)), sorted(set(se.tolist()))))
    for k in (2, 4, 8):
        tab = np.zeros((k, 8), dtype=np.int64)
        for s_, e_ in zip(slot % k, se):
            tab[s_, e_] += 1
        print('  (blockIdx >> 3) %% %d against the shader engine:' % k, tab[:, :max(se) + 1].tolist())
    for e_ in sorted(set(se.tolist())):
        m = se == e_
        print('  shader engine %d: %d workgroups, mean lifetime %.0f' % (e_, m.sum(), (t_out - t_in)[m].mean()))
    sys.exit(0)
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
