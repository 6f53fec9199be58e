"""CPU: host logic of the drop-in classes -- state_dict layout, checkpoint round trip with the
oracle, loud failure without a HIP device. No compute."""
import json
import os

import pytest
import torch

from oracle import pianobart_oracle as O
from tests.golden_util import GOLD, load_vocab, randomize_params

E2W, W2E = load_vocab()


def _cfgs(S, d, L, f, h):
    from pianobart_amd.model import BartConfig
    kw = dict(max_position_embeddings=S, d_model=d, encoder_layers=L, decoder_layers=L, encoder_ffn_dim=f,
              decoder_ffn_dim=f, encoder_attention_heads=h, decoder_attention_heads=h)
    return BartConfig(**kw), O.BartConfig(**kw)


def test_state_dict_layout_and_roundtrip():
    from pianobart_amd.model import PianoBart, PianoBartLM
    g = json.load(open(os.path.join(GOLD, 'g9_state_dict.json')))
    c, oc = _cfgs(128, 128, 2, 512, 4)
    m = PianoBartLM(PianoBart(c, E2W, W2E))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g['cfg1']
    assert sum(p.numel() for p in m.parameters()) == g['cfg1_n_params']
    o = O.PianoBartLM(O.PianoBart(oc, E2W, W2E))
    randomize_params(o, 3)
    m.load_state_dict(o.state_dict(), strict=True)                 # reference-format checkpoint loads strictly
    o2 = O.PianoBartLM(O.PianoBart(oc, E2W, W2E))
    o2.load_state_dict(m.state_dict(), strict=True)                # and what we save loads back into the reference layout
    for (k1, v1), (k2, v2) in zip(o.state_dict().items(), o2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    # pre-train checkpoints hold PianoBart only (pretrain.py:96-110): 105 tensors at 2 layers
    assert len(m.pianobart.state_dict()) == 105
    assert m.pianobart.decoder_linear is m.pianobart.encoder_linear
    assert m.pianobart.bart.encoder.embed_tokens.weight is m.pianobart.bart.shared.weight


def test_public_attributes():
    from pianobart_amd.model import PianoBart
    c, _ = _cfgs(32, 64, 1, 64, 4)
    pb = PianoBart(c, E2W, W2E)
    assert pb.n_tokens == [262, 134, 135, 262, 134, 38, 260, 55] and pb.hidden_size == 64 and pb.bar_pad_word == 256
    assert list(pb.pad_word_np) == [256, 128, 129, 256, 128, 32, 254, 49]
    assert list(pb.sos_word_np) == [258, 130, 131, 258, 130, 34, 256, 51]
    assert list(pb.mask_word_np - pb.pad_word_np) == [1] * 8 and list(pb.eos_word_np - pb.pad_word_np) == [3] * 8
    t = pb.get_rand_tok()
    assert t.shape == (8,) and all(0 <= t[i] < pb.n_tokens[i] for i in range(8))


def test_no_cpu_execution_path():
    from pianobart_amd._lib import PBError
    from pianobart_amd.model import PianoBart, PianoBartLM
    c, _ = _cfgs(32, 64, 1, 64, 4)
    m = PianoBartLM(PianoBart(c, E2W, W2E)).eval()
    ids = torch.zeros(1, 32, 8, dtype=torch.long)
    with pytest.raises(PBError):
        m(ids, ids, torch.ones(1, 32), torch.ones(1, 32))
    with pytest.raises(PBError):
        m.pianobart(ids, None, torch.ones(1, 32), None)


def test_engine_flat_layout_covers_every_live_parameter():
    from pianobart_amd.engine import Engine
    from pianobart_amd.model import PianoBart, PianoBartLM
    c, _ = _cfgs(32, 64, 2, 128, 4)
    m = PianoBartLM(PianoBart(c, E2W, W2E))
    eng = Engine(m.pianobart, m.mask_lm, 'fp32')
    pm = eng._param_map()
    live = {id(p) for n, p in m.named_parameters() if 'shared' not in n and 'embed_tokens' not in n}
    assert {id(p) for p, _, _ in pm} == live and len(pm) == len(live)
    spans = sorted((eng._elem_off(s, r), eng._elem_off(s, r) + p.numel()) for p, s, r in pm)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))      # no overlap
    assert spans[-1][1] <= eng.n_total


def test_pretrain_cli_defaults_match_reference():
    """pretrain.py:22-44 defaults."""
    from pianobart_amd.pretrain import get_args_pretrain
    a = get_args_pretrain([])
    assert (a.dict_file, a.name, a.num_workers, a.batch_size, a.mask_percent, a.max_seq_len) == ('./Data/Octuple.pkl', 'pianobart', 5, 16, 0.15, 1024)
    assert (a.hs, a.layers, a.ffn_dims, a.heads, a.epochs, a.lr, a.cpu) == (1024, 8, 2048, 8, 500, 2e-5, False)
    assert a.datasets == ['asap', 'EMOPIA', 'Pianist8', 'POP1K7', 'POP909']


def test_int16_shards_roundtrip(tmp_path):
    import numpy as np
    from pianobart_amd.data import MidiDataset, convert_to_int16
    from tests.golden_util import synth_octuple_batch
    a = synth_octuple_batch(5, 32, seed=1)[5].numpy()
    np.save(tmp_path / 'a.npy', a)
    assert convert_to_int16(str(tmp_path / 'a.npy'), str(tmp_path / 'a16.npy')) == (5, 32, 8)
    ds = MidiDataset(str(tmp_path / 'a16.npy'))
    assert len(ds) == 5 and ds[3].dtype == torch.int16 and np.array_equal(ds[3].numpy(), a[3])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))
    assert batch.shape == (2, 32, 8)


def test_fast_host_sampler_reproduces_sampling_draw_for_draw():
    import numpy as np
    import torch
    """PianoBartLM.sample_row (the generate loop's host sampler) must give the tokens of the reference-faithful sampling() / nucleus()
    (model.py:84-107, pinned by golden G7) AND leave the global np.random stream in the same state, for flat, peaked and tied logits."""
    from pianobart_amd import ops
    from pianobart_amd.model import PianoBartLM, sampling
    rng = np.random.default_rng(5)
    n_multi = 0
    for trial in range(1200):
        scale = [0.05, 1.0, 4.0, 12.0][trial % 4]
        row = rng.normal(scale=scale, size=ops.VOCAB).astype(np.float32)
        if trial % 7 == 0:
            row = np.round(row)                                   # ties
        np.random.seed(trial)
        ref = [int(sampling(torch.from_numpy(row[ops.SEG_OFF[j]:ops.SEG_OFF[j + 1]].copy()), PianoBartLM.SAMPLE_P[j], PianoBartLM.SAMPLE_T[j]))
               for j in range(8)]
        st_ref = np.random.get_state()
        np.random.seed(trial)
        got = PianoBartLM.sample_row(PianoBartLM, torch.from_numpy(row.copy())).tolist()
        st_got = np.random.get_state()
        assert got == ref, (trial, got, ref)
        assert st_ref[2] == st_got[2] and np.array_equal(st_ref[1], st_got[1])
        n_multi += int(ref[3] != int(np.argmax(row[ops.SEG_OFF[3]:ops.SEG_OFF[4]])))
    assert n_multi > 20          # the p = 0.9 heads really sampled (not just argmax) in a good share of the trials


def test_module_prefixed_checkpoint_loads(capsys):
    """Fine-tune checkpoints saved under nn.DataParallel carry `module.` keys (finetune_generation.py:276-285); the reference's demo.py:128-129
    loads them with strict=False, i.e. loads NOTHING. The drop-in strips the prefix, and says so; a file with no matching key is reported."""
    from pianobart_amd.model import PianoBart, PianoBartLM, checkpoint_state_dict
    c, oc = _cfgs(32, 64, 1, 64, 4)
    o = O.PianoBartLM(O.PianoBart(oc, E2W, W2E))
    randomize_params(o, 11)
    wrapped = torch.nn.DataParallel(o).state_dict()                              # what the reference writes from a multi-GPU fine-tune
    assert all(k.startswith('module.') for k in wrapped)
    m = PianoBartLM(PianoBart(c, E2W, W2E))
    res = m.load_state_dict(checkpoint_state_dict(wrapped, m), strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    for (k1, v1), (k2, v2) in zip(o.state_dict().items(), m.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert "stripping the 'module.' prefix" in capsys.readouterr().out
    plain = o.state_dict()
    assert checkpoint_state_dict(plain, m) is plain and capsys.readouterr().out == ''
    checkpoint_state_dict(o.pianobart.state_dict(), m)                           # a pre-train file (PianoBart only) into PianoBartLM: the reference's silent no-op
    assert 'WARNING: none of the checkpoint' in capsys.readouterr().out
