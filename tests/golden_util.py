"""Shared, deterministic test helpers: seeded non-trivial weights and synthetic Octuple batches.

Used by oracle/make_goldens.py (to produce the committed vectors) and by the tests (to
re-create the identical inputs/weights). Any change here invalidates tests/golden/*.npz:
the sha256 of the state_dict stored in the fixtures catches that.
"""
import hashlib
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')

PAD = np.array([256, 128, 129, 256, 128, 32, 254, 49], dtype=np.int64)   # classes order
N_TOK = [262, 134, 135, 262, 134, 38, 260, 55]
HI = [255, 127, 128, 127, 127, 31, 253, 48]                               # max "real" id per column


def load_vocab():
    with open(os.path.join(ROOT, 'pianobart_amd', 'data', 'octuple_vocab.json')) as f:
        e2w = json.load(f)['e2w']
    w2e = {k: {v: w for w, v in d.items()} for k, d in e2w.items()}
    return e2w, w2e


def sd_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def randomize_params(model, seed):
    """Seeded, *non-trivial* parameters (non-zero biases, non-unit LayerNorm, weights large
    enough that attention is far from uniform) so that parity tests exercise every term."""
    g = torch.Generator().manual_seed(seed)
    seen = set()
    for name, p in model.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        with torch.no_grad():
            if 'layer_norm' in name or 'layernorm' in name:
                if name.endswith('weight'):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif 'lut' in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif 'shared' in name or 'embed_tokens' in name:
                p.zero_()                                    # dead table (never read in forward)
            elif 'embed_positions' in name:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif name.endswith('bias'):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif 'encoder_linear' in name or 'decoder_linear' in name:
                p.copy_(torch.randn(p.shape, generator=g) / (16.0 * np.sqrt(p.shape[1])) * 2.0)
            else:
                fan_in = p.shape[-1]
                p.copy_(torch.randn(p.shape, generator=g) * (1.6 / np.sqrt(fan_in)))


def synth_octuple_batch(B, S, seed, min_len=None):
    """Synthetic Octuple sequences per SURVEY 8(d): L~U{S/2..S}; bar non-decreasing; EOS row;
    PAD tail. Returns (enc_ids, dec_ids, loss_mask, enc_mask, dec_mask, target) where enc_ids is
    a TokenMask-style corruption of target (80/10/10) and loss_mask ~ Bernoulli(0.15) over all S."""
    rng = np.random.default_rng(seed)
    tgt = np.zeros((B, S, 8), dtype=np.int64)
    for b in range(B):
        L = int(rng.integers(min_len if min_len is not None else S // 2, S + 1))
        L = max(2, min(L, S))
        rows = np.stack([rng.integers(0, HI[c] + 1, size=L - 1) for c in range(8)], axis=1)
        rows[:, 0] = np.minimum(np.cumsum(rng.random(L - 1) < 0.06), 255)
        tgt[b, :L - 1] = rows
        tgt[b, L - 1] = PAD + 3                                  # EOS row
        tgt[b, L:] = PAD
    sel = rng.random((B, S)) < 0.15
    sel[:, 0] |= ~sel.any(axis=1)                                # at least one masked position / sample
    kind = rng.random((B, S))
    enc = tgt.copy()
    mask_row = PAD + 1
    for b in range(B):
        for s in np.nonzero(sel[b])[0]:
            if kind[b, s] < 0.8:
                enc[b, s] = mask_row
            elif kind[b, s] < 0.9:
                enc[b, s] = [rng.integers(0, N_TOK[c]) for c in range(8)]
    dec = np.zeros_like(tgt)
    dec[:, 1:] = tgt[:, :-1]
    dec[:, 0] = PAD + 2                                          # SOS row
    loss_mask = np.repeat(sel[:, :, None], 8, axis=2).astype(np.float32)
    emask = (enc[:, :, 0] != 256).astype(np.float32)
    dmask = (dec[:, :, 0] != 256).astype(np.float32)
    t = torch.from_numpy
    return t(enc), t(dec), t(loss_mask), t(emask), t(dmask), t(tgt)


# ---------------------------------------------------------------- gen_mask decision replay (SURVEY a-9)
def g6_replay_cases():
    """The 25 single-sequence cases (choices 1-5 x seeds 0-4) and the 3-sample batch of tests/golden/g6_gen_mask.npz, each with the
    random DECISIONS the oracle drew under the golden's seeds (oracle Corruptor.trace; layout: include/pianobart_hip.h,
    pb_corrupt_replay). Returns (single, batch): lists of dicts with ids (S,8) int64, choice, dec int32, rand_rows (S,8) int16 or
    None, masked (S,8) int64 and pos (S,) int64 = the reference's outputs."""
    import random
    from oracle import pianobart_oracle as O
    e2w, w2e = load_vocab()
    z = np.load(os.path.join(GOLD, 'g6_gen_mask.npz'))
    pb = O.PianoBart(O.BartConfig(max_position_embeddings=64, d_model=32, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=64,
                                  decoder_ffn_dim=64, encoder_attention_heads=4, decoder_attention_heads=4), e2w, w2e)
    corr = O.Corruptor(pb, 64, 0.15)
    ids = torch.from_numpy(z['ids']).long()

    def case(ids_row, masked, pos):
        tr = corr.trace
        return dict(ids=ids_row.numpy().astype(np.int64), choice=tr['choice'], dec=tr['dec'], rand_rows=tr.get('rand_rows'),
                    masked=np.asarray(masked).astype(np.int64), pos=np.asarray(pos).astype(np.int64).reshape(len(ids_row), -1)[:, 0])

    single = []
    for choice in range(1, 6):
        for seed in range(5):
            random.seed(seed); np.random.seed(seed)
            corr.trace = {}
            masked, pos = corr.gen_mask(ids.clone(), choice)
            c = case(ids, z['masked_c%d_s%d' % (choice, seed)], z['pos_c%d_s%d' % (choice, seed)])
            assert np.array_equal(np.asarray(masked).astype(np.int64), c['masked'])          # oracle == reference (G6)
            single.append(c)
    batch = []
    random.seed(7); np.random.seed(7)
    ori = torch.from_numpy(z['batch']).long()
    for b in range(ori.shape[0]):                                   # pretrain.py:127-153 draws the choice, then the corruption, per sample
        corr.trace = {}
        corr.gen_mask(ori[b].clone())
        batch.append(case(ori[b], z['batch_enc'][b], z['batch_loss_mask'][b][:, 0]))
    corr.trace = None
    return single, batch


def apply_decisions_numpy(c, mask_percent=0.15):
    """Host restatement of the APPLY stage of pb_corrupt (csrc/pb_corrupt.hip) on one case of g6_replay_cases(): decisions + input ->
    (out (S,8), loss mask (S,)). Shows that the decision layout carries all of gen_mask's randomness (CPU test); the kernel itself is
    checked on the GPU."""
    ids, dec, ch = c['ids'], c['dec'], c['choice']
    S = ids.shape[0]
    mask_row = PAD + 1
    out = np.zeros_like(ids); lm = np.zeros(S, dtype=np.int64)
    if ch == 1:
        k = int(S * mask_percent)
        dele = dec[:S] != 0
        first = int(np.argmax(dele)) if dele.any() else S
        kept = ids[~dele]
        out[:len(kept)] = kept; out[S - k:] = PAD
        lm[first:] = 1
    elif ch == 2:
        out = ids.copy()
        out[dec[:S] == 1] = mask_row
        out[dec[:S] == 2] = c['rand_rows'][dec[:S] == 2]
        lm = (dec[:S] != 0).astype(np.int64)
    elif ch == 3:
        order = sorted(range(S), key=lambda i: (dec[ids[i, 0]], i))
        out = ids[order]
        lm = (out != ids).any(1).astype(np.int64)
    elif ch == 4:
        ok = False
        for att in range(10):
            rows, i, s = [], 0, 0
            while i < S:
                v = dec[att * S + s]; s += 1
                if v == 0:
                    rows += [ids[i], mask_row]; i += 1
                elif v > 0:
                    rows.append(mask_row); i += v
                else:
                    rows.append(ids[i]); i += 1
            if len(rows) <= S:
                rows += [PAD] * (S - len(rows)); ok = True
                break
        out = np.stack(rows) if ok else ids.copy()
        lm = (out != ids).any(1).astype(np.int64) if ok else np.zeros(S, dtype=np.int64)
    else:
        ran = int(dec[0])
        out = np.roll(ids, -ran, axis=0)
        lm[:] = 1 if ran != 0 else 0
    return out, lm
