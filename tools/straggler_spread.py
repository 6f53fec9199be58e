#!/usr/bin/env python3
"""Per-rank step-time spread of the packed pre-train step (VERDICT r2 #7b): with dead-row compaction a rank's step time depends on ITS
samples' padding, and under data parallelism the ranks meet at every bucket exchange, so the slowest rank sets the pace. One GPU,
sequentially: the bench model, 32 different SURVEY 8(d) batches (seed = 1234 + "rank"), median of 5 steps each. Prints the spread
and what sorting a global batch by kept rows before dealing it to the ranks would leave of it."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pianobart_amd import ops
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, synth_octuple_batch
    e2w, w2e = load_vocab()
    dev = torch.device('cuda', 0)
    cfg = BartConfig(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
                     encoder_attention_heads=12, decoder_attention_heads=12, dropout=0.1)
    torch.manual_seed(0)
    m = PianoBartLM(PianoBart(cfg, e2w, w2e, precision='bf16')).train().to(dev)
    eng = m._get_engine()
    eng.bind(dev)
    eng.pipeline_updates = True
    nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    res = []
    for r in range(nranks):
        enc, dec, loss_mask, emask, dmask, target = [t.to(dev) for t in synth_octuple_batch(32, 1024, seed=1234 + r)]
        args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
        ts = []
        for it in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.loss_and_grads(*args, train=True)
            eng.optimizer_step(lr=2e-5)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res.append(dict(rank=r, ms=float(np.median(ts[2:])), rows=eng.last_rows, pairs=eng.last_pairs))
        print(res[-1], flush=True)
    ms = np.array([x['ms'] for x in res])
    rows = np.array([x['rows'][0] + x['rows'][1] for x in res])
    groups = [ms[i:i + 8] for i in range(0, len(ms) - 7, 8)]
    out = dict(n=len(ms), ms_min=float(ms.min()), ms_median=float(np.median(ms)), ms_max=float(ms.max()),
               spread_pct=float((ms.max() - ms.min()) / np.median(ms) * 100),
               straggler_cost_pct_8ranks=float(np.mean([(g.max() - g.mean()) / g.mean() * 100 for g in groups])) if groups else None,
               corr_ms_vs_rows=float(np.corrcoef(ms, rows)[0, 1]), rows_min=int(rows.min()), rows_max=int(rows.max()))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
