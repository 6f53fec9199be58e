"""CPU: the oracle (oracle/pianobart_oracle.py) against the vectors captured from the real
reference by oracle/make_goldens.py. This is what pins the oracle."""
import json
import os
import random

import numpy as np
import torch

from oracle import pianobart_oracle as O
from tests.golden_util import GOLD, load_vocab, randomize_params, sd_checksum, synth_octuple_batch

E2W, W2E = load_vocab()


def _cfg(S, d, L, f, h, dropout=0.1):
    return O.BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=L, decoder_layers=L,
                        encoder_ffn_dim=f, decoder_ffn_dim=f, encoder_attention_heads=h,
                        decoder_attention_heads=h, dropout=dropout)


def _lm(S, d, L, f, h, seed, dropout=0.1):
    m = O.PianoBartLM(O.PianoBart(_cfg(S, d, L, f, h, dropout), E2W, W2E))
    randomize_params(m, seed)
    return m


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def _sha(z):
    return bytes(z['sd_sha']).decode()


def test_vocab_layout():
    assert list(E2W.keys()) == ['Bar', 'Position', 'Pitch', 'Duration', 'Velocity', 'Instrument', 'Tempo', 'TimeSig']
    assert [len(E2W[k]) for k in O.CLASSES] == [262, 134, 135, 262, 134, 38, 260, 55]
    assert O.loss_weights(E2W) == [262, 134, 262, 134, 38, 135, 55, 260]      # dict order (SURVEY a-8)


def test_g1_forward_loss_argmax():
    z = np.load(os.path.join(GOLD, 'g1_forward_cfg1.npz'))
    m = _lm(128, 128, 2, 512, 4, seed=11).eval()
    assert sd_checksum(m.state_dict()) == _sha(z)
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(2, 128, seed=5)
    assert np.array_equal(enc.numpy(), z['enc']) and np.array_equal(dec.numpy(), z['dec'])
    dmask = torch.from_numpy(z['dmask'])
    with torch.no_grad():
        y = m(enc, dec, emask, dmask)
        h = m.pianobart(enc, dec, emask, dmask)
        e = m.pianobart(enc, None, emask, None)
    logits = torch.cat(y, dim=-1)
    # tolerance: fp32 restatement vs fp32 reference, implementation noise only (3e-7 sdpa vs eager)
    assert _rel(logits, torch.from_numpy(z['logits'])) < 2e-5
    assert _rel(h.last_hidden_state, torch.from_numpy(z['hidden'])) < 2e-5
    assert _rel(h.encoder_last_hidden_state, torch.from_numpy(z['enc_hidden'])) < 2e-5
    assert _rel(e.last_hidden_state, torch.from_numpy(z['enc_only_hidden'])) < 2e-5
    total, losses, accs, arg = O.pretrain_loss(y, target, loss_mask, E2W)
    assert abs(float(total) - float(z['total_loss'])) < 1e-5
    assert np.allclose(torch.stack(losses).numpy(), z['head_losses'], atol=1e-5)
    assert np.allclose(torch.stack(accs).numpy(), z['head_acc'], atol=1e-6)
    assert np.array_equal(arg.numpy(), z['argmax'])


def test_g4_grads_and_adamw():
    z = np.load(os.path.join(GOLD, 'g4_grads_small.npz'))
    m = _lm(64, 64, 2, 128, 4, seed=23, dropout=0.0).train()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(2, 64, seed=9)
    y = m(enc, dec, emask, dmask)
    total, *_ = O.pretrain_loss(y, target, loss_mask, E2W)
    total.backward()
    assert abs(float(total) - float(z['total_loss'])) < 1e-5
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads.keys()) == list(z['param_names'])
    for k in z.files:
        if k.startswith('grad__'):
            assert _rel(grads[k[6:]], torch.from_numpy(z[k])) < 5e-4, k
    norms = torch.stack([grads[k].double().norm().float() for k in z['param_names']])
    assert np.allclose(norms.numpy(), z['per_param_grad_norm'], rtol=2e-3, atol=1e-6)
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    assert abs(float(gn) - float(z['grad_norm'])) / float(z['grad_norm']) < 1e-4


def test_g6_gen_mask_exact():
    z = np.load(os.path.join(GOLD, 'g6_gen_mask.npz'))
    pb = O.PianoBart(_cfg(64, 32, 1, 64, 4), E2W, W2E)
    corr = O.Corruptor(pb, 64, 0.15)
    ids = torch.from_numpy(z['ids']).long()
    for choice in range(1, 6):
        for seed in range(5):
            random.seed(seed); np.random.seed(seed)
            masked, pos = corr.gen_mask(ids.clone(), choice)
            assert np.array_equal(np.asarray(masked).astype(np.int64), z['masked_c%d_s%d' % (choice, seed)].astype(np.int64)), (choice, seed)
            assert np.array_equal(np.asarray(pos).astype(np.int64), z['pos_c%d_s%d' % (choice, seed)].astype(np.int64)), (choice, seed)
    random.seed(7); np.random.seed(7)
    enc, dec, lm, em, dm = O.pretrain_batch(corr, torch.from_numpy(z['batch']).long())
    assert np.array_equal(enc.numpy(), z['batch_enc']) and np.array_equal(dec.numpy(), z['batch_dec'])
    assert np.array_equal(lm.numpy().astype(np.uint8), z['batch_loss_mask'])
    assert np.array_equal(em.numpy(), z['batch_emask']) and np.array_equal(dm.numpy(), z['batch_dmask'])


def test_g6_gen_mask_invariants():
    """Structure tests modelled on the reference's own hand-made gen_mask inputs (pretrain.py:582-688)."""
    pb = O.PianoBart(_cfg(16, 32, 1, 64, 4), E2W, W2E)
    corr = O.Corruptor(pb, 16, 0.5)
    ids = torch.tensor([[8 * i + c for c in range(8)] for i in range(16)]) % 30
    ids[:, 0] = torch.arange(16) // 4
    random.seed(1); np.random.seed(1)
    masked, pos = corr.gen_mask(ids.clone(), 1)            # deletion: PAD count == deleted count
    assert int((masked[:, 0] == 256).sum()) == int(16 * 0.5) and masked.shape == ids.shape
    masked, pos = corr.gen_mask(ids.clone(), 3)            # permutation keeps the row multiset
    assert sorted(map(tuple, masked.tolist())) == sorted(map(tuple, ids.tolist()))
    masked, pos = corr.gen_mask(ids.clone(), 5)            # rotation keeps the row multiset
    assert sorted(map(tuple, masked.tolist())) == sorted(map(tuple, ids.tolist()))
    masked, pos = corr.gen_mask(ids.clone(), 2)
    assert int(pos.sum()) == round(16 * 0.5)


def test_g7_sampling():
    z = np.load(os.path.join(GOLD, 'g7_sampling.npz'))
    logits = torch.from_numpy(z['logits'])
    picks = []
    np.random.seed(2023)
    for i in range(6):
        for p, t in ((1, 1.2), (0.9, 1.0), (0.9, 2.0)):
            picks.append(O.sampling(logits[i], p, t))
    assert picks == list(z['picks']) and np.random.rand() == float(z['rng_after'])
    # p = 1 => always argmax (SURVEY a-6)
    for i in range(6):
        assert picks[3 * i] == int(logits[i].argmax())


def test_g8_generate_trace():
    z = np.load(os.path.join(GOLD, 'g8_generate.npz'))
    m = _lm(24, 64, 2, 128, 4, seed=31).eval()
    assert sd_checksum(m.state_dict()) == _sha(z)
    enc = torch.from_numpy(z['enc']).long(); emask = torch.from_numpy(z['emask'])
    with torch.no_grad():
        np.random.seed(2023)
        out = m(enc, None, emask, None, generate=True)
    assert np.array_equal(out.numpy(), z['tokens'])


def test_g9_state_dict_layout():
    g = json.load(open(os.path.join(GOLD, 'g9_state_dict.json')))
    m = _lm(128, 128, 2, 512, 4, seed=0)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g['cfg1']
    assert sum(p.numel() for p in m.parameters()) == g['cfg1_n_params']
    with torch.device('meta'):
        big = O.PianoBartLM(O.PianoBart(_cfg(1024, 768, 12, 3072, 12), E2W, W2E))
    assert [[k, list(v.shape)] for k, v in big.state_dict().items()] == g['cfg2']
    assert sum(p.numel() for p in big.parameters()) == g['cfg2_n_params']


def test_hf_adamw_formula():
    """HF AdamW (4.29.2) restatement vs an independent scalar derivation (SURVEY a-11)."""
    p = torch.tensor([1.0, -2.0]); g = torch.tensor([0.5, 0.25])
    m = torch.zeros(2); v = torch.zeros(2)
    O.hf_adamw_step([p], [g], [m], [v], step=1, lr=0.1)
    m1 = 0.1 * g; v1 = 0.001 * g * g
    upd = (0.1 * (1 - 0.999) ** 0.5 / (1 - 0.9)) * m1 / (v1.sqrt() + 1e-6)
    exp = torch.tensor([1.0, -2.0]) - upd
    exp = exp - 0.1 * 0.01 * exp
    assert torch.allclose(p, exp, atol=1e-7)


def test_g6_decision_trace_carries_all_randomness():
    """The decisions recorded by the oracle (the replay input of pb_corrupt_replay) + the deterministic APPLY stage reproduce the
    reference's gen_mask outputs for all 25 golden cases and the golden batch."""
    from tests.golden_util import apply_decisions_numpy, g6_replay_cases
    single, batch = g6_replay_cases()
    assert sorted(set(c['choice'] for c in single)) == [1, 2, 3, 4, 5] and len(single) == 25 and len(batch) == 3
    for c in single + batch:
        out, lm = apply_decisions_numpy(c)
        assert np.array_equal(out, c['masked']) and np.array_equal(lm, c['pos']), c['choice']


def test_g14_generation_trainer_step():
    """G14 = the real reference GenerationTrainer (finetune_generation.py:118-272) run by oracle/make_goldens.py: loss, per-head CE,
    accuracies, argmax ids in test mode; gradient norm (what its clip_grad_norm_ reports) and named gradients of a train-mode batch."""
    z = np.load(os.path.join(GOLD, 'g14_generation_trainer.npz'))
    m = O.PianoBartLM(O.PianoBart(_cfg(64, 64, 1, 128, 4, dropout=0.0), E2W, W2E))
    randomize_params(m, 13)
    assert sd_checksum(m.state_dict()) == str(z['sd'])
    x, y = torch.from_numpy(z['x']).long(), torch.from_numpy(z['y']).long()
    with torch.no_grad():
        total, ce, accs, arg = O.generation_step(m.eval(), x, y, E2W)
    assert abs(round(float(total), 4) - float(z['test_loss'])) < 1.5e-4
    assert np.allclose([float(c) for c in ce], z['test_head_ce'], rtol=2e-5)
    assert np.allclose([round(float(a), 4) for a in accs], z['test_accs'], atol=1.01e-4)
    assert np.array_equal(arg.numpy(), z['test_all_output'].astype(np.int64))
    # the line the reference prints carries the WEIGHTED per-head losses (0.3 / 1.5 / 1)
    printed = [float(v) for v in str(z['test_stdout'][0]).split('loss:')[1].split(',')]
    assert np.allclose(printed, np.array([float(c) for c in ce]) * np.array(O.GENERATION_HEAD_WEIGHT), atol=2e-6 + 2e-5 * max(printed))
    m.train()
    total, ce, accs, arg = O.generation_step(m, x, y, E2W)
    total.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    assert abs(gn - float(z['train_gnorm'])) < 5e-4 * gn
    for i, k in enumerate(z['train_grad_names']):
        assert _rel(grads[str(k)], torch.from_numpy(z['train_grad_%d' % i])) < 5e-4, k
