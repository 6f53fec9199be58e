"""GPU: the pre-train harness (counterpart of pretrain.py / main.pretrain) end to end on cfg-1 shape."""
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_util import ROOT, load_vocab, synth_octuple_batch

pytestmark = pytest.mark.gpu
E2W, W2E = load_vocab()
VOCAB_JSON = os.path.join(ROOT, 'pianobart_amd', 'data', 'octuple_vocab.json')


def _write_dataset(root, n=12, S=128):
    os.makedirs(os.path.join(root, 'syn'), exist_ok=True)
    seqs = synth_octuple_batch(n, S, seed=77)[5].numpy().astype(np.int64)
    for name, part in (('train', seqs[:8]), ('test', seqs[8:10]), ('valid', seqs[10:])):
        np.save(os.path.join(root, 'syn', 'syn_%s_split.npy' % name), part)


def test_main_pretrain_cfg1_end_to_end(tmp_path, capsys):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from oracle import pianobart_oracle as O
    from pianobart_amd.pretrain import pretrain
    data_root = str(tmp_path / 'Data' / 'output_pretrain')
    _write_dataset(data_root)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        np.random.seed(0); torch.manual_seed(0)
        tr = pretrain(['--dict_file', VOCAB_JSON, '--name', 't', '--datasets', 'syn', '--num_workers', '0', '--batch_size', '2',
                       '--max_seq_len', '128', '--hs', '128', '--layers', '2', '--ffn_dims', '512', '--heads', '4', '--epochs', '3',
                       '--lr', '1e-3', '--cuda_devices', '0', '--precision', 'fp32', '--data_root', data_root])
        out = capsys.readouterr().out
        log = open('result/pretrain/t/log').read().splitlines()
        ck = torch.load('result/pretrain/t/model.ckpt', weights_only=False)
        best_exists = os.path.exists('result/pretrain/t/model_best.ckpt')
    finally:
        os.chdir(cwd)
    # log + stdout formats of main.py:84-92 and pretrain.py:199-204
    assert len(log) == 4 and all(re.match(r'Epoch \d+: train_loss=[\d.]+, train_acc=\[.*\], valid_loss=[\d.]+, valid_acc=\[.*\]$', l) for l in log[:3])
    assert log[3].startswith('Time cost in pretrain of PianoBart is ')
    assert re.search(r'^Loss: \d+\.\d{6} \| loss: ' + ', '.join([r'\d+\.\d{6}'] * 8) + '$', out, re.M)
    assert re.search(r'^Acc: \d+\.\d{6} \| acc: ', out, re.M) and 'epoch: 3/3 | Train Loss: ' in out
    # checkpoint layout of pretrain.py:96-110
    assert set(ck.keys()) == {'epoch', 'state_dict', 'best_acc', 'valid_acc', 'valid_loss', 'train_loss', 'optimizer'}
    assert ck['epoch'] == 3 and len(ck['state_dict']) == 105 and best_exists
    cfg = O.BartConfig(max_position_embeddings=128, d_model=128, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512,
                       decoder_ffn_dim=512, encoder_attention_heads=4, decoder_attention_heads=4)
    O.PianoBart(cfg, E2W, W2E).load_state_dict(ck['state_dict'], strict=True)        # loads into the reference layout
    # golden artefacts of the REFERENCE's own main.pretrain() run (oracle/make_goldens.py g11): same structure, number for number
    import json
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_pretrain_artifacts.json')))
    shape = lambda line: re.sub(r'[-+]?\d+\.?\d*(?:e[-+]?\d+)?', '#', line)
    assert sorted(ck.keys()) == g['ckpt_keys'] and list(ck['state_dict'].keys()) == g['state_dict_keys']
    assert shape(log[0]) == shape(g['log'].splitlines()[0])
    first = lambda pre: [l for l in out.splitlines() if l.startswith(pre)][0]
    assert shape(first('Loss: ')) == shape(g['stdout_loss_line']) and shape(first('Acc: ')) == shape(g['stdout_acc_line'])
    assert shape(first('epoch: ')) == shape(g['stdout_epoch_line'])
    losses = [float(re.search(r'train_loss=([\d.]+)', l).group(1)) for l in log[:3]]
    assert losses[-1] < losses[0]                                                       # it learns


def test_main_pretrain_in_the_split_bf16_instantiation(tmp_path, capsys):
    """`--precision bf16x3` through the whole driver (device corruption, fused step with split-bf16 GEMMs and fused split-bf16 attention, clip, AdamW,
    validation, checkpoint): the run learns, its checkpoint loads into the reference layout, and its first-epoch losses are the exact-f32 run's to 1e-3
    (same seeds: the two instantiations draw the same corruption and dropout bits)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from oracle import pianobart_oracle as O
    from pianobart_amd.pretrain import pretrain
    data_root = str(tmp_path / 'Data' / 'output_pretrain')
    _write_dataset(data_root)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    logs = {}
    try:
        for prec in ('bf16x3', 'fp32'):
            np.random.seed(0); torch.manual_seed(0)
            import random
            random.seed(0)
            pretrain(['--dict_file', VOCAB_JSON, '--name', prec, '--datasets', 'syn', '--num_workers', '0', '--batch_size', '2', '--max_seq_len', '128',
                      '--hs', '128', '--layers', '2', '--ffn_dims', '512', '--heads', '4', '--epochs', '2', '--lr', '1e-3', '--cuda_devices', '0',
                      '--precision', prec, '--data_root', data_root, '--quiet'])
            logs[prec] = open('result/pretrain/%s/log' % prec).read().splitlines()
        ck = torch.load('result/pretrain/bf16x3/model.ckpt', weights_only=False)
    finally:
        os.chdir(cwd)
    capsys.readouterr()
    cfg = O.BartConfig(max_position_embeddings=128, d_model=128, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512,
                       decoder_ffn_dim=512, encoder_attention_heads=4, decoder_attention_heads=4)
    O.PianoBart(cfg, E2W, W2E).load_state_dict(ck['state_dict'], strict=True)
    loss = lambda l: float(re.search(r'train_loss=([\d.]+)', l).group(1))
    x3, f32 = [loss(l) for l in logs['bf16x3'][:2]], [loss(l) for l in logs['fp32'][:2]]
    assert x3[1] < x3[0]
    assert abs(x3[0] - f32[0]) <= 2e-3 * f32[0] + 1.01e-3, (x3, f32)          # the log rounds to 3 decimals


def test_pretrain_resume_continues_the_run(tmp_path, capsys):
    """--resume (the checkpoint's moments had no loader and the heads' moments no weights beside them): the file holds PianoBart's
    state_dict (reference format), and inside 'optimizer' the LM heads + named AdamW moments + step; a resumed run continues epoch numbering,
    step count and log, and a fresh Pretrainer that resumes holds bit for bit what was saved. A reference-written optimizer state is refused."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from pianobart_amd._lib import PBError
    from pianobart_amd.model import BartConfig, PianoBart
    from pianobart_amd.pretrain import Pretrainer, pretrain
    data_root = str(tmp_path / 'Data' / 'output_pretrain')
    _write_dataset(data_root)
    common = ['--dict_file', VOCAB_JSON, '--name', 'r', '--datasets', 'syn', '--num_workers', '0', '--batch_size', '2', '--max_seq_len', '128',
              '--hs', '64', '--layers', '1', '--ffn_dims', '128', '--heads', '4', '--lr', '1e-3', '--cuda_devices', '0', '--data_root', data_root, '--quiet']
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        pretrain(common + ['--epochs', '1'])
        ck1 = torch.load('result/pretrain/r/model.ckpt', weights_only=False)
        os.replace('result/pretrain/r/model.ckpt', 'first.ckpt')
        capsys.readouterr()
        pretrain(common + ['--epochs', '3', '--resume', 'first.ckpt'])
        out = capsys.readouterr().out
        ck3 = torch.load('result/pretrain/r/model.ckpt', weights_only=False)
        log = open('result/pretrain/r/log').read().splitlines()
        kw = dict(max_position_embeddings=128, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=128, decoder_ffn_dim=128,
                  encoder_attention_heads=4, decoder_attention_heads=4)
        tr = Pretrainer(PianoBart(BartConfig(**kw), E2W, W2E), None, None, 1e-3, 2, 128, 0.15, False, [0])
        assert tr.resume('first.ckpt') == (1, ck1['best_acc'])
        st = tr.engine.optimizer_state(tr.model)
        ref_style = dict(ck1, optimizer={'state': {}, 'param_groups': [{'lr': 1e-3}]})
        torch.save(ref_style, 'ref_style.ckpt')
        with pytest.raises(PBError):
            tr.resume('ref_style.ckpt')
    finally:
        os.chdir(cwd)
    assert set(ck1.keys()) == {'epoch', 'state_dict', 'best_acc', 'valid_acc', 'valid_loss', 'train_loss', 'optimizer'}      # still the reference's keys
    nb = ck1['optimizer']['step']
    assert ck1['epoch'] == 1 and nb >= 3 and ck3['epoch'] == 3 and ck3['optimizer']['step'] == 3 * nb
    assert 'resumed from first.ckpt at epoch 1' in out and 'epoch: 2/3' in out and 'epoch: 3/3' in out and 'epoch: 1/3' not in out
    assert [l.split(':')[0] for l in log if l.startswith('Epoch')] == ['Epoch 1', 'Epoch 2', 'Epoch 3']
    assert st['step'] == nb and set(st['exp_avg']) == set(ck1['optimizer']['exp_avg'])
    for k, v in ck1['optimizer']['exp_avg_sq'].items():
        assert torch.equal(st['exp_avg_sq'][k], v) and torch.equal(st['exp_avg'][k], ck1['optimizer']['exp_avg'][k]), k
    for k, v in ck1['optimizer']['mask_lm'].items():
        assert torch.equal(tr.model.mask_lm.state_dict()[k].cpu(), v), k
    for k, v in ck1['state_dict'].items():
        assert torch.equal(tr.pianobart.state_dict()[k].cpu(), v), k
    assert any(not torch.equal(ck3['state_dict'][k], ck1['state_dict'][k]) for k in ck1['state_dict'] if 'fc1.weight' in k)     # and it kept training


def test_pretrainer_step_matches_oracle_on_its_own_batch():
    """One Pretrainer batch (device corruption + shift-right + masks) -> fused loss == oracle loss on the same tensors."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from oracle import pianobart_oracle as O
    from pianobart_amd.model import BartConfig, PianoBart
    from pianobart_amd.pretrain import Pretrainer
    from tests.golden_util import randomize_params
    kw = dict(max_position_embeddings=64, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=128,
              decoder_ffn_dim=128, encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0)
    pb = PianoBart(BartConfig(**kw), E2W, W2E, precision='fp32')
    tr = Pretrainer(pb, None, None, 2e-5, 4, 64, 0.15, False, [0])
    randomize_params(tr.model, 3)
    tr.engine.bind(tr.device)
    batch = synth_octuple_batch(4, 64, seed=8)[5]
    enc16, dec16, tgt16, lm, em, dm = tr.prepare_batch(batch)
    # decoder input = shift-right with SOS (pretrain.py:132-139); masks = bar column != 256 (:151-153)
    assert torch.equal(dec16.long().cpu(), O.shift_right(batch, pb.sos_word_np))
    assert torch.equal(em.cpu(), (enc16[:, :, 0].long().cpu() != 256).float())
    sums = tr.engine.loss_and_grads(enc16, dec16, tgt16, lm, em, dm, train=False).double().cpu()
    w = torch.tensor(O.loss_weights(E2W), dtype=torch.double)
    mine = float(((sums[0:8] / sums[8:16]) * w).sum() / w.sum())
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**kw), E2W, W2E)).eval()
    o.load_state_dict({k: v.cpu() for k, v in tr.model.state_dict().items()}, strict=True)
    with torch.no_grad():
        total, *_ = O.pretrain_loss(o(enc16.long().cpu(), dec16.long().cpu(), em.cpu(), dm.cpu()), batch, lm.cpu(), E2W)
    assert abs(mine - float(total)) / float(total) < 1e-4
    masked, pos = tr.gen_mask(batch[0], 2)
    assert masked.shape == (64, 8) and pos.shape == (64,) and int(pos.sum()) == round(64 * 0.15)
    bad = batch.clone(); bad[2, 5, 4] = 134                               # Duration table has 134 rows: nn.Embedding's IndexError (PianoBart.py:15-16)
    with pytest.raises(IndexError):
        tr.prepare_batch(bad)


def test_generation_trainer_step_matches_the_reference_trainer_g14(capsys):
    """GenerationTrainer against G14 = the REAL reference GenerationTrainer (finetune_generation.py:118-272, `y_shift = x` :155, head
    weights :239-250) run on the CPU by oracle/make_goldens.py: test-mode loss / accuracies / argmax ids / printed lines, and the
    gradients of a train-mode batch (what its clip_grad_norm_ reports). No formula restated here."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import os
    from pianobart_amd.finetune_generation import GenerationTrainer
    from pianobart_amd.model import BartConfig, PianoBart
    from tests.golden_util import GOLD, randomize_params, sd_checksum
    z = np.load(os.path.join(GOLD, 'g14_generation_trainer.npz'))
    kw = dict(max_position_embeddings=64, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=128,
              decoder_ffn_dim=128, encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0)
    x, y = torch.from_numpy(z['x']).long(), torch.from_numpy(z['y']).long()
    tr = GenerationTrainer(PianoBart(BartConfig(**kw), E2W, W2E, precision='fp32'), [(x, y)], [(x, y)], [(x, y)], 1e-3, (4, 64, 8), False, [0])
    randomize_params(tr.model, 13)
    assert sd_checksum({k: v.cpu() for k, v in tr.model.state_dict().items()}) == str(z['sd'])
    tr.engine.bind(tr.device)
    capsys.readouterr()
    loss, accs, fb, fa, all_out = tr.test()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith(('Loss:', 'Acc:'))]
    assert abs(loss - float(z['test_loss'])) < 1.5e-4 and np.allclose(accs, z['test_accs'], atol=1.01e-4)
    assert torch.equal(all_out.long(), torch.from_numpy(z['test_all_output'].astype(np.int64)))
    num = lambda line: np.array([float(v) for v in line.replace('|', ',').replace('Loss:', '').replace('Acc:', '').replace('loss:', '').replace('acc:', '').split(',')])
    for mine, ref in zip(lines, z['test_stdout']):                        # "Loss: total | loss: 8 weighted heads", "Acc: mean | acc: 8 heads"
        assert mine.split()[0] == str(ref).split()[0]
        assert np.allclose(num(mine), num(str(ref)), atol=3e-6, rtol=1e-4), (mine, str(ref))
    assert tr.valid()[:2] == (loss, accs)
    # train mode, one batch: the gradients the update is made from (before the clip), against the reference's
    tloss, taccs = tr.train()[:2]
    assert abs(tloss - float(z['train_loss'])) < 1.5e-4 and np.allclose(taccs, z['train_accs'], atol=1.01e-4)
    eng = tr.engine
    views = eng.grad_views_of(eng.Gcur)
    by_id = {id(p): g for p, g in zip(eng.params, views)}
    named = {k: by_id[id(p)] for k, p in tr.model.named_parameters() if id(p) in by_id}
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in views)))
    assert abs(gn - float(z['train_gnorm'])) < 1e-3 * gn
    for i, k in enumerate(z['train_grad_names']):
        ref = torch.from_numpy(z['train_grad_%d' % i])
        got = named[str(k)].float().cpu()
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-3, k
    l1 = tr.train()[0]
    assert l1 < tloss                                                     # the update went downhill


def test_optimizer_state_is_keyed_by_parameter_name_and_resumes_bit_for_bit():
    """the AdamW moments travel keyed by state_dict names (the flat layout is an implementation detail): two steps, save, one more
    step -- against a FRESH model that loads the weights and the named moments and takes the same third step: identical parameters. A state
    with flat buffers (older commits) or foreign names is refused."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from pianobart_amd import ops
    from pianobart_amd._lib import PBError
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import randomize_params
    kw = dict(max_position_embeddings=64, d_model=64, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=128,
              decoder_ffn_dim=128, encoder_attention_heads=1, decoder_attention_heads=1, dropout=0.0)
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(2, 64, seed=3)]
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)

    def fresh():
        m = PianoBartLM(PianoBart(BartConfig(**kw), E2W, W2E, precision='bf16'))
        randomize_params(m, 5)
        m = m.train().cuda()
        eng = m._get_engine(); eng.bind(torch.device('cuda', 0))
        return m, eng

    def step(eng):
        eng.loss_and_grads(*args, train=True, ids_checked=True)
        eng.optimizer_step(lr=1e-3)
    m, eng = fresh()
    step(eng); step(eng)
    st = eng.optimizer_state(m)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    assert st['step'] == 2 and set(st['exp_avg']) == {k for k, _ in m.named_parameters() if not k.endswith('shared.weight')}
    assert all(st['exp_avg'][k].shape == p.shape for k, p in m.named_parameters() if k in st['exp_avg'])
    step(eng); torch.cuda.synchronize()
    want = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m2, eng2 = fresh()
    m2.load_state_dict(sd, strict=True)
    eng2.bind(torch.device('cuda', 0)); eng2.refresh_shadow(force=True)
    eng2.load_optimizer_state(m2, st)
    step(eng2); torch.cuda.synchronize()
    got = m2.state_dict()
    for k in want:
        assert torch.equal(want[k], got[k].cpu()), k
    with pytest.raises(PBError):
        eng2.load_optimizer_state(m2, {'step': 2, 'exp_avg': torch.zeros(8), 'exp_avg_sq': torch.zeros(8)})
    bad = dict(st, exp_avg={('x.' + k): v for k, v in st['exp_avg'].items()})
    with pytest.raises(PBError):
        eng2.load_optimizer_state(m2, bad)


def test_mmap_int16_shard_feeds_the_same_batch_as_the_int64_path(tmp_path):
    """a memory-mapped int16 shard -> DataLoader -> Pretrainer.prepare_batch gives bit-identical device tensors to the
    reference-format int64 array path (same corruption seed), and the ids reach the device as int16 without a conversion kernel."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from torch.utils.data import DataLoader
    from pianobart_amd.data import MidiDataset, convert_to_int16
    from pianobart_amd.model import BartConfig, PianoBart
    from pianobart_amd.pretrain import Pretrainer
    seqs = synth_octuple_batch(6, 64, seed=12)[5].numpy().astype(np.int64)
    np.save(tmp_path / 'a.npy', seqs)
    convert_to_int16(str(tmp_path / 'a.npy'), str(tmp_path / 'b.npy'))
    kw = dict(max_position_embeddings=64, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=128,
              decoder_ffn_dim=128, encoder_attention_heads=4, decoder_attention_heads=4)
    tr = Pretrainer(PianoBart(BartConfig(**kw), E2W, W2E, precision='fp32'), None, None, 2e-5, 6, 64, 0.15, False, [0])
    b16 = next(iter(DataLoader(MidiDataset(str(tmp_path / 'b.npy')), batch_size=6)))
    assert b16.dtype == torch.int16
    import random
    seed0 = tr._step_seed
    random.seed(3)
    got = tr.prepare_batch(b16)
    tr._step_seed = seed0
    random.seed(3)
    want = tr.prepare_batch(torch.from_numpy(seqs))                              # the reference's int64 rows
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and torch.equal(a, b)
    assert torch.equal(got[2].cpu().long(), torch.from_numpy(seqs))              # targets = the sequences themselves


def test_demo_midi_in_generate_midi_out(tmp_path):
    """demo.py:105-170 end to end on the device: .mid -> Midi2Octuple -> checkpoint loaded with strict=False -> model(generate=True)
    -> Octuple2Midi -> .mid. 2-layer shape, window 32: the checkpoint's head biases make the special ids unsamplable, so all 32
    positions are generated (a random-init model would stop at once); the written file reads back as the generated rows."""
    from pianobart_amd import demo as D, octuple_midi as om
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params
    e2w, w2e = load_vocab()
    S = 32
    song = om.Song(480, [(120 * i, 120 * i + 200, 48 + (7 * i) % 24, 64 + i % 32, 0, False) for i in range(20)], [(0, 4, 4)], [(0, 120.0)])
    src = str(tmp_path / 'in.mid')
    om.write_midi(song, src)
    kw = dict(max_position_embeddings=S, d_model=128, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
              encoder_attention_heads=2, decoder_attention_heads=2)
    m = PianoBartLM(PianoBart(BartConfig(**kw), e2w, w2e))
    randomize_params(m, 5)
    with torch.no_grad():
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            m.mask_lm.proj[i].bias[p0:] = -30.0
        m.mask_lm.proj[2].bias[1:] = -30.0                               # piano only (program 0): the written notes share one track
        m.mask_lm.proj[3].bias[128:] = -30.0                             # melodic pitches
    ckpt = str(tmp_path / 'lm.ckpt')
    torch.save({'state_dict': m.state_dict()}, ckpt)
    out = str(tmp_path / 'out.mid')
    np.random.seed(2023)
    args = D.Args(ckpt=ckpt, input=src, output=out, max_seq_len=S, hs=128, layers=2, ffn_dims=256, heads=2)
    x, y = D.demo(args)
    assert tuple(x.shape) == (1, S, 8) and tuple(y.shape) == (1, S, 8) and y.is_cuda and y.dtype == torch.int64
    assert [tuple(r) for r in x[0, :20, :4].tolist()] == [r[:4] for r in om.midi_to_encoding(song)]
    rows = om.octuple_to_rows(y.cpu().numpy())
    assert len(rows) == S - 1 and int((y[0, :, 0] < 256).sum()) == S      # every position was generated; the last row becomes the EOS (demo.py:88-89)
    back = om.read_midi(out)
    assert len(back.notes) == len(om.encoding_to_midi(rows).notes) > 0
    # the same call through the command-line surface, without a checkpoint: nothing but the flags differs
    a2 = D.get_args(['--nopretrain', '--input', src, '--output', str(tmp_path / 'o2.mid'), '--max_seq_len', str(S), '--hs', '128', '--layers', '2',
                     '--ffn_dims', '256', '--heads', '2'])
    x2, y2 = D.demo(a2)
    assert torch.equal(x2, x) and tuple(y2.shape) == (1, S, 8)
    with pytest.raises(Exception):
        D.demo(D.Args(cpu=True, nopretrain=True, input=src))
