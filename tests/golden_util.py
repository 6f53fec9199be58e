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
