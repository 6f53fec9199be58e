"""One-XCD persistent decode kernel (PB_DECODE_GRAPH=2, pb_decode1.hip) against the graph of 6 launches per layer (=1): the same forced
tokens, the logits row of every step, and the per-token time.   python tools/decode1_check.py [--small]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import engine as E
from tests.golden_util import synth_octuple_batch
from tests.test_model_gpu import _lm

shapes = [(200, 256, 2, 512, 4), (96, 256, 2, 512, 2), (72, 512, 2, 512, 8), (40, 1024, 2, 512, 8), (130, 768, 2, 3072, 12)] if '--small' in sys.argv else \
         [(130, 768, 2, 3072, 12), (1024, 768, 12, 3072, 12)]
for S, d, L, ffn, heads in shapes:
    m = _lm(S, d, L, ffn, heads, 31, 'bf16').eval()
    with torch.no_grad():
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            m.mask_lm.proj[i].bias[p0:] = -30.0
    m = m.cuda()
    enc = synth_octuple_batch(1, S, seed=8, min_len=max(S - 9, S * 2 // 3))[5].cuda()
    emask = (enc[:, :, 0] != 256).float()
    N = min(S, 200)
    forced = synth_octuple_batch(1, S, seed=23, min_len=S)[5][0]
    forced[-1] = forced[-2]
    eng = m._get_engine()

    def run(mode):
        E._DECODE_GRAPH = mode
        rows = []

        def feed(row):
            rows.append(row.clone())
            if len(rows) > N:
                return torch.tensor([256, 128, 129, 256, 128, 32, 254, 49])
            return forced[len(rows) - 1].clone()
        eng.generate(enc, emask, feed)
        return rows, eng.last_decode
    g, ig = run(1)
    line = 'S=%d d=%d L=%d ffn=%d H=%d: graph %.3f ms/token (%d launches)' % (S, d, L, ffn, heads, ig['loop_ms'] / ig['tokens'], ig['launches_per_token'])
    for mode, name in ((2, 'one XCD'), (3, 'all XCDs')):
        o, io = run(mode)
        o2, io2 = run(mode)
        worst = 0.0
        for i in range(min(len(g), len(o))):
            keep = g[i] > -20
            worst = max(worst, float((o[i][keep] - g[i][keep]).abs().max() / g[i][keep].abs().max()))
        same = all(torch.equal(a, b) for a, b in zip(o, o2))
        line += ' | %s: %.3f / %.3f ms/token (%d launch), rows %d/%d, worst logits rel %.2e, repeatable %s' % (
            name, io['loop_ms'] / io['tokens'], io2['loop_ms'] / io2['tokens'], io['launches_per_token'], len(o), len(g), worst, same)
    print(line, flush=True)
