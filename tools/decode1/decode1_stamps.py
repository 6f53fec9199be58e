"""Cycle stamps of the one-XCD decode kernel (diagnostic build: PB_EXTRA_HIPCC_FLAGS=-DPB_D1_STAMPS python -m pianobart_amd.build --force).
Per barrier three stamps: phase end, requests issued + workgroup synchronised, barrier passed. Prints, per phase of the last token, the
compute time, the wait inside the barrier, for participants 0 and 17.   python tools/decode1_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pianobart_amd import engine as E
from pianobart_amd._lib import LIB
from tests.golden_util import synth_octuple_batch
from tests.test_model_gpu import _lm
S, d, L, ffn, heads = 1024, 768, 12, 3072, 12
m = _lm(S, d, L, ffn, heads, 31, 'bf16').eval()
with torch.no_grad():
    for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
        m.mask_lm.proj[i].bias[p0:] = -30.0
m = m.cuda()
enc = synth_octuple_batch(1, S, seed=8, min_len=700)[5].cuda()
emask = (enc[:, :, 0] != 256).float()
forced = synth_octuple_batch(1, S, seed=23, min_len=S)[5][0]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = []
def feed(row):
    rows.append(1)
    return forced[len(rows) - 1].clone() if len(rows) <= N else torch.tensor([256, 128, 129, 256, 128, 32, 254, 49])
E._DECODE_GRAPH = int(os.environ.get('MODE', '2'))
m._get_engine().generate(enc, emask, feed)
info = m._get_engine().last_decode
print(info)
st = np.zeros(1024, dtype=np.uint64)
lib = LIB.load()
assert lib.pb_decode1_stamps(ctypes.c_void_p(st.ctypes.data)) == 0
names = ['1 qkv', '2 self-attn', '3 out', '4 LN1+q_c', '5 cross-attn', '6 out_c', '7 LNc+fc1', '8 fc2']
for w in range(2):
    s = st[w * 512:(w + 1) * 512].astype(np.int64)
    n = int((s != 0).sum())
    t0 = s[0]
    TK = float(os.environ.get('TICKS_PER_US', '100'))
    print('participant %d: %d stamps, token = %d ticks (at %.0f ticks/us: %.1f us; the loop measured %.1f us/token incl. host)' % (0 if w == 0 else 17, n, s[n - 1] - t0, TK, (s[n - 1] - t0) / TK, 1e3 * info['loop_ms'] / info['tokens']))
    # stamps: [0] start, then per barrier (phase end, synced, passed), last = end
    nb = (n - 2) // 3
    tot = np.zeros((8, 3))
    for b in range(nb):
        prev = s[3 * (b - 1) + 3] if b else s[0]
        e, sy, ps = s[1 + 3 * b], s[2 + 3 * b], s[3 + 3 * b]
        tot[b % 8] += (e - prev, sy - e, ps - sy)
    for k in range(8):
        print('   phase %-14s work %7.2f us   store-wait + requests + sync %6.2f us   barrier wait %6.2f us   (per layer, mean)' % (names[k], *(tot[k] / (nb / 8) / TK)))
    print('   sum per layer %.2f us; head phase %.2f us' % (tot.sum() / (nb / 8) / TK, (s[n - 1] - s[n - 2]) / TK))

sub = np.zeros(32, dtype=np.uint64)
assert lib.pb_decode1_sub(ctypes.c_void_p(sub.ctypes.data), 1) == 0
labels = {0: 'barriers (store wait, requests, sync, poll)', 1: 'before gemv (LN / ctx / row tail)', 2: 'gemv: sync xs', 3: 'gemv: units', 4: 'gemv: sync partials', 5: 'gemv: rows out',
          6: 'attn: K/V requests', 7: 'attn: q row in', 8: 'attn: sync', 9: 'attn: items', 10: 'attn: sync', 11: 'attn: merge + records out', 12: 'ctx: records in (sc1)',
          13: 'ctx: merge', 14: 'LN: rows in (sc1)', 15: 'LN: statistics + out', 16: 'row in (sc1)'}
tok = info['tokens'] + 0.0
tot = float(sub.sum())
print('participant 0, ticks per token by sub-phase (sum %.0f):' % (tot / tok))
for k in range(17):
    print('   %-46s %9.0f  %5.1f %%' % (labels[k], sub[k] / tok, 100.0 * sub[k] / tot))
