"""GPU: device corruption kernel (counterpart of Pretrainer.gen_mask, pretrain.py:211-546).
Bit parity: the kernel's DECIDE stage is replaced by the reference's own random decisions (recorded by the oracle under the golden
seeds) and its outputs must equal tests/golden/g6_gen_mask.npz exactly (test_replay_*). The Philox-driven DECIDE stage cannot
reproduce the Mersenne-Twister streams of Python's `random`; for it the structure the reference's own hand-made gen_mask inputs
are about (pretrain.py:582-688) is asserted exactly, plus the distributions."""
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
        # the loss mask starts at the first deleted index (identical neighbouring rows -- the PAD tail -- make the greedy match late)
        assert first <= removed[0] and (ids[b, first:removed[0] + 1] == ids[b, first]).all()
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


def _replay(ops, cases, p=0.15):
    from pianobart_amd._lib import LIB
    B, S = len(cases), cases[0]['ids'].shape[0]
    stride = int(LIB.query('pb_corrupt_replay_stride', S))
    dec = np.zeros((B, stride), dtype=np.int32)
    rr = np.zeros((B, S, 8), dtype=np.int16)
    for b, c in enumerate(cases):
        dec[b, :len(c['dec'])] = c['dec']
        if c['rand_rows'] is not None:
            rr[b] = c['rand_rows']
    ids16 = ops.ids_to_i16(torch.from_numpy(np.stack([c['ids'] for c in cases])).cuda())
    out = torch.full_like(ids16, -1); lm = torch.full((B, S, 8), -1.0, device='cuda')
    ch = torch.tensor([c['choice'] for c in cases], dtype=torch.int32, device='cuda')
    ops.corrupt_replay(ids16, out, lm, ch, p, torch.from_numpy(dec).cuda(), torch.from_numpy(rr).cuda(), PAD, MASK)
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.int64), lm.cpu().numpy()


def test_replay_matches_reference_goldens_bit_exact(ops):
    """SURVEY a-9: with the reference's random decisions (seeds 0-4 x choices 1-5 of G6) the kernel's outputs and loss masks are
    the reference's, bit for bit: 25 cases in one launch."""
    from tests.golden_util import g6_replay_cases
    single, _ = g6_replay_cases()
    out, lm = _replay(ops, single)
    for b, c in enumerate(single):
        assert np.array_equal(out[b], c['masked']), ('rows', c['choice'], b % 5)
        assert np.array_equal(lm[b], np.repeat(c['pos'][:, None], 8, 1).astype(np.float32)), ('loss mask', c['choice'], b % 5)


def test_replay_batch_construction_matches_reference(ops):
    """pretrain.py:127-153 on the golden 3-sample batch (choices drawn by random.randint under seed 7): corrupted encoder ids, loss mask,
    shift-right decoder ids and both attention masks equal the reference's."""
    from tests.golden_util import GOLD, g6_replay_cases
    import os
    z = np.load(os.path.join(GOLD, 'g6_gen_mask.npz'))
    _, batch = g6_replay_cases()
    out, lm = _replay(ops, batch)
    assert np.array_equal(out, z['batch_enc'].astype(np.int64))
    assert np.array_equal(lm.astype(np.uint8), z['batch_loss_mask'])
    tgt16 = ops.ids_to_i16(torch.from_numpy(z['batch'].astype(np.int64)).cuda())
    dec16 = torch.empty_like(tgt16)
    ops.shift_right(tgt16, torch.tensor(PAD + 2, dtype=torch.int16, device='cuda'), dec16, 3, 64)
    assert np.array_equal(dec16.cpu().numpy(), z['batch_dec'])
    assert np.array_equal((out[:, :, 0] != 256).astype(np.float32), z['batch_emask'])
    assert np.array_equal((dec16[:, :, 0] != 256).float().cpu().numpy(), z['batch_dmask'])


def test_replay_against_the_oracle_on_random_and_degenerate_sequences(ops):
    """Differential run beyond the 25 golden cases: 5 corruptions x 48 sequences -- random Octuple rows of every length from one EOS
    row to a full window, single-bar pieces, pieces whose every row opens a new bar, all-PAD windows -- each corrupted by the oracle's
    restatement of pretrain.py:211-546 under its own seed, its random decisions replayed through pb_corrupt_replay: rows and loss mask
    bit for bit."""
    import random
    from oracle import pianobart_oracle as O
    from tests.golden_util import load_vocab
    e2w, w2e = load_vocab()
    S = 64
    pb = O.PianoBart(O.BartConfig(max_position_embeddings=S, d_model=32, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=64,
                                  decoder_ffn_dim=64, encoder_attention_heads=4, decoder_attention_heads=4), e2w, w2e)
    corr = O.Corruptor(pb, S, 0.15)
    rng = np.random.default_rng(99)
    hi = np.array([256, 128, 129, 256, 128, 32, 254, 49])
    pad, eos = pb.pad_word_np.astype(np.int64), pb.eos_word_np.astype(np.int64)

    def piece(L, bars):
        x = np.tile(pad, (S, 1))
        if L > 0:
            body = rng.integers(0, hi, size=(L, 8))
            if bars == 'one':
                body[:, 0] = 3
            elif bars == 'every':
                body[:, 0] = np.arange(L) % 256
            else:
                body[:, 0] = np.minimum(np.cumsum(rng.random(L) < 0.2), 255)
            x[:L] = body
            x[L - 1] = eos
        return x
    lengths = [0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 63, 64]
    seqs = [piece(L, b) for L in lengths for b in ('walk', 'one', 'every')] + [piece(int(rng.integers(2, S + 1)), 'walk') for _ in range(12)]
    cases = []
    for k, x in enumerate(seqs):
        for choice in range(1, 6):
            random.seed(1000 + 7 * k + choice); np.random.seed(1000 + 7 * k + choice)
            corr.trace = {}
            masked, pos = corr.gen_mask(torch.from_numpy(x).long(), choice)
            tr = corr.trace
            cases.append(dict(ids=x.astype(np.int64), choice=tr['choice'], dec=tr['dec'], rand_rows=tr.get('rand_rows'),
                              masked=np.asarray(masked).astype(np.int64), pos=np.asarray(pos).astype(np.int64).reshape(S, -1)[:, 0]))
    corr.trace = None
    for lo in range(0, len(cases), 60):
        chunk = cases[lo:lo + 60]
        out, lm = _replay(ops, chunk)
        for b, c in enumerate(chunk):
            assert np.array_equal(out[b], c['masked']), ('rows', c['choice'], lo + b)
            assert np.array_equal(lm[b], np.repeat(c['pos'][:, None], 8, 1).astype(np.float32)), ('loss mask', c['choice'], lo + b)


def test_philox_and_replay_share_the_apply_stage(ops, batch):
    """Decisions read back from a Philox run (what was deleted / masked / where rows went) and replayed give the identical output:
    the two DECIDE sources feed one APPLY stage."""
    ids = batch.numpy()
    B, S = ids.shape[:2]
    out1, lm1, _ = _run(ops, batch, 1, 5)
    cases = []
    for b in range(B):
        # recover the deleted flags of the Philox run: greedy subsequence match, ambiguous only among identical neighbouring rows
        k = int(S * 0.15); kept = out1[b, :S - k]; j = 0; dec = np.zeros(S, dtype=np.int32)
        for i in range(S):
            if j < S - k and (ids[b, i] == kept[j]).all():
                j += 1
            else:
                dec[i] = 1
        cases.append(dict(ids=ids[b], choice=1, dec=dec, rand_rows=None))
    out2, lm2 = _replay(ops, cases)
    assert np.array_equal(out1, out2)
    out5, lm5, _ = _run(ops, batch, 5, 13)
    cases = [dict(ids=ids[b], choice=5, dec=np.array([[r for r in range(S) if (np.roll(ids[b], -r, axis=0) == out5[b]).all()][0]], dtype=np.int32),
                  rand_rows=None) for b in range(B)]
    out6, lm6 = _replay(ops, cases)
    assert np.array_equal(out5, out6) and np.array_equal(lm5, lm6)
