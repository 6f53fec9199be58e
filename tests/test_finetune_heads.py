"""fine-tune heads (SequenceClassification, TokenClassification incl. the velocity task's decoder label-embedding swap)
against G12, the logits / loss / gradients of the REAL reference (oracle/make_goldens.py g12). CPU: the oracle restatement is pinned to
G12. GPU: the product path (HIP ops through autograd) is compared with G12 in both precisions, and FinetuneTrainer runs end to end."""
import os
import numpy as np
import pytest
import torch

from tests.golden_util import load_vocab, randomize_params, sd_checksum, synth_octuple_batch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
E2W, W2E = load_vocab()
S, D, L, F, H = 64, 128, 2, 256, 4
KW = dict(max_position_embeddings=S, d_model=D, encoder_layers=L, decoder_layers=L, encoder_ffn_dim=F, decoder_ffn_dim=F,
          encoder_attention_heads=H, decoder_attention_heads=H, dropout=0.0)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _build(mod, cfgcls, pbcls, tag, **pbkw):
    pb = pbcls(cfgcls(**KW), E2W, W2E, **pbkw)
    m = {'seq': lambda: mod.SequenceClassification(pb, 8, D), 'tok4': lambda: mod.TokenClassification(pb, 4, D),
         'tok8': lambda: mod.TokenClassification(pb, 8, D)}[tag]()
    randomize_params(m, 31)
    for x in m.modules():
        if isinstance(x, torch.nn.Dropout):
            x.p = 0.0
    return m.train()


def _run(m, z, tag, dev, lossf):
    enc = torch.from_numpy(z['enc']).long().to(dev); emask = torch.from_numpy(z['emask']).to(dev)
    y = torch.from_numpy(z[tag + '_y']).to(dev)
    if tag == 'seq':
        yh = m(input_ids_encoder=enc, encoder_attention_mask=emask)
        return yh, lossf(yh, y, None, True)
    if tag == 'tok4':
        yh = m(input_ids_encoder=enc, input_ids_decoder=enc, encoder_attention_mask=emask, decoder_attention_mask=emask)
    else:
        yh = m(input_ids_encoder=enc, input_ids_decoder=torch.from_numpy(z['tok8_y_shift']).to(dev), encoder_attention_mask=emask,
               decoder_attention_mask=torch.from_numpy(z['tok8_attn_shift']).to(dev))
    return yh, lossf(yh, y, emask, False)


def _check(m, z, tag, yh, loss, tol_logits, tol_grad):
    assert _rel(yh.detach(), torch.from_numpy(z[tag + '_logits'])) < tol_logits
    assert abs(float(loss) - float(z[tag + '_loss'])) < max(tol_logits, 1e-5) * abs(float(z[tag + '_loss'])) * 10
    m.zero_grad()
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    for i, k in enumerate(z[tag + '_grad_names']):
        r = _rel(grads[str(k)], torch.from_numpy(z['%s_grad_%d' % (tag, i)]))
        assert r < tol_grad, (tag, str(k), r)
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    assert abs(gn - float(z[tag + '_gnorm'])) < tol_grad * float(z[tag + '_gnorm'])


@pytest.mark.parametrize('tag', ['seq', 'tok4', 'tok8'])
def test_oracle_heads_match_reference_golden(tag):
    from oracle import pianobart_oracle as O
    z = np.load(os.path.join(GOLD, 'g12_finetune_heads.npz'))
    m = _build(O, O.BartConfig, O.PianoBart, tag)
    assert sd_checksum(m.state_dict()) == str(z[tag + '_sd'])         # same weights as the reference run
    yh, loss = _run(m, z, tag, 'cpu', O.finetune_loss)
    _check(m, z, tag, yh, loss, 2e-5, 5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol_logits,tol_grad', [('fp32', 1e-4, 2e-3), ('bf16x3', 1e-4, 2e-3), ('bf16', 6e-2, 2e-1)])
@pytest.mark.parametrize('tag', ['seq', 'tok4', 'tok8'])
def test_heads_match_reference_golden(tag, precision, tol_logits, tol_grad):
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    from pianobart_amd import model as M
    from pianobart_amd.finetune import FinetuneTrainer
    z = np.load(os.path.join(GOLD, 'g12_finetune_heads.npz'))
    m = _build(M, M.BartConfig, M.PianoBart, tag, precision=precision)
    assert sd_checksum(m.state_dict()) == str(z[tag + '_sd'])         # state_dict keys / shapes / values identical to the reference's
    m = m.cuda()
    lossf = lambda yh, y, mask, seq: FinetuneTrainer.compute_loss(None, yh, y, mask, seq)
    yh, loss = _run(m, z, tag, 'cuda', lossf)
    _check(m, z, tag, yh, loss, tol_logits, tol_grad)


@pytest.mark.gpu
@pytest.mark.parametrize('task', ['composer', 'velocity'])
def test_finetune_trainer_end_to_end(task):
    """FinetuneTrainer train / valid / test on a synthetic set: runs, learns (the loss of a memorisable 8-sample set drops),
    returns the reference's tuple shapes, and steps backbone AND head parameters."""
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    from torch.utils.data import DataLoader
    from pianobart_amd import model as M
    from pianobart_amd.finetune import FinetuneDataset, FinetuneTrainer
    torch.manual_seed(0)
    X = synth_octuple_batch(8, S, seed=3)[0].numpy()
    seq = task == 'composer'
    rng = np.random.default_rng(0)
    y = rng.integers(0, 8, size=(8,)) if seq else rng.integers(0, 7, size=(8, S))
    mk = lambda: DataLoader(FinetuneDataset(X, y), batch_size=4)
    pb = M.PianoBart(M.BartConfig(**dict(KW, dropout=0.1)), E2W, W2E, precision='fp32')
    tr = FinetuneTrainer(pb, mk(), mk(), mk(), lr=1e-3, class_num=8 if seq else 7, hs=D, testset_shape=y.shape, cpu=False, cuda_devices=[0], SeqClass=seq)
    head_before = [p.detach().clone() for p in tr.head_optim.params]
    bb_before = pb.bart.encoder.layers[0].fc1.weight.detach().clone()
    l0, a0 = tr.train()
    for _ in range(5):
        l1, a1 = tr.train()
    vl, va = tr.valid()
    tl, ta, out = tr.test()
    assert np.isfinite([l0, l1, vl, tl]).all() and l1 < l0
    assert tuple(out.shape) == tuple(y.shape) and 0.0 <= ta <= 1.0
    assert all(not torch.equal(a, b.detach()) for a, b in zip(head_before, tr.head_optim.params))
    assert not torch.equal(bb_before, pb.bart.encoder.layers[0].fc1.weight.detach())


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['seq', 'tok4'])
def test_weight_regulariser_matches_oracle(tag):
    """--weight (finetune.py:241-243): penalty value and the gradients it adds (per parameter TENSOR: q / k / v separately, zero-norm
    tensors get 0) against the oracle's autograd on the same weights and inputs."""
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    from torch.utils.data import DataLoader
    from oracle import pianobart_oracle as O
    from pianobart_amd import model as M
    from pianobart_amd.finetune import FinetuneDataset, FinetuneTrainer
    z = np.load(os.path.join(GOLD, 'g12_finetune_heads.npz'))
    W = 0.01
    mo = _build(O, O.BartConfig, O.PianoBart, tag)
    yo, lo = _run(mo, z, tag, 'cpu', O.finetune_loss)
    live = [p for k, p in mo.named_parameters() if 'shared' not in k and 'embed_tokens' not in k]
    pen_all = float(O.l2_penalty(list(mo.parameters()), W))
    mo.zero_grad()
    (lo + O.l2_penalty(live, W)).backward()
    ref = {k: p.grad.clone() for k, p in mo.named_parameters() if p.grad is not None}

    m = _build(M, M.BartConfig, M.PianoBart, tag, precision='fp32').cuda()
    seq = tag == 'seq'
    dl = DataLoader(FinetuneDataset(z['enc'], z[tag + '_y']), batch_size=z['enc'].shape[0])
    tr = FinetuneTrainer(m.pianobart, dl, dl, dl, lr=1e-3, class_num=8 if seq else 3, hs=D, testset_shape=z[tag + '_y'].shape, cpu=False,
                         cuda_devices=[0], model=m, SeqClass=seq, weight=W)
    lossf = lambda yh, y, mask, s: tr.compute_loss(yh, y, mask, s)
    yh, loss = _run(m, z, tag, 'cuda', lossf)
    m.zero_grad()
    loss.backward()
    pen = float(tr.l2_penalty(True))
    assert abs(pen - pen_all) < 1e-5 * pen_all
    eng = tr.engine
    slot = {id(p): i for i, p in enumerate(eng.params)}
    n = 0
    for k, p in m.named_parameters():
        if k not in ref:
            continue
        g = eng.grad_views[slot[id(p)]] if id(p) in slot else p.grad
        r = _rel(g.detach().cpu(), ref[k])
        assert r < 2e-3, (k, r)
        n += 1
    assert n > 40
    # evaluation mode: value only, gradients untouched
    before = eng.G32.clone()
    assert abs(float(tr.l2_penalty(False)) - pen_all) < 1e-5 * pen_all
    assert torch.equal(before, eng.G32)


def _write_finetune_sets(root, prefix, seq, gen=False):
    X = synth_octuple_batch(8, S, seed=3)[5].numpy().astype(np.int64)
    rng = np.random.default_rng(0)
    for part, sl in (('train', slice(0, 4)), ('valid', slice(4, 6)), ('test', slice(6, 8))):
        np.save(os.path.join(root, '%s_%s.npy' % (prefix, part)), X[sl])
        y = X[sl] if gen else (rng.integers(0, 8, size=(sl.stop - sl.start,)) if seq else rng.integers(0, 4, size=(sl.stop - sl.start, S)))
        np.save(os.path.join(root, '%s_%s_ans.npy' % (prefix, part)), y)


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_finetune_driver_writes_one_log_line_per_epoch(tmp_path, precision):
    """finetune() (main.py:103-215) end to end on a tiny synthetic composer set: the log holds the header line and ONE line per
    epoch (real newlines), checkpoints carry the reference's keys."""
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    import pickle
    from pianobart_amd.finetune import finetune
    root = str(tmp_path / 'data')
    os.makedirs(root)
    _write_finetune_sets(root, 'Pianist8', True)
    pickle.dump((E2W, W2E), open(tmp_path / 'dict.pkl', 'wb'))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        finetune(['--task', 'composer', '--dataset', 'Pianist8', '--dataroot', root, '--dict_file', str(tmp_path / 'dict.pkl'), '--name', 't',
                  '--num_workers', '0', '--batch_size', '2', '--max_seq_len', str(S), '--hs', str(D), '--layers', '1', '--ffn_dims', '128',
                  '--heads', '4', '--epochs', '2', '--nopretrain', '--cuda_devices', '0', '--precision', precision])
        log = open('result/finetune/composer_t/log').read()
        ck = torch.load('result/finetune/composer_t/model.ckpt', weights_only=False)
    finally:
        os.chdir(cwd)
    lines = log.split('\n')
    assert '\\n' not in log and lines[0].startswith('Loading pre-trained model from') and lines[-1] == ''
    assert len(lines) == 4 and all(l.startswith('Epoch %d: train_loss=' % (i + 1)) for i, l in enumerate(lines[1:3]))
    assert set(ck.keys()) == {'epoch', 'state_dict', 'valid_acc', 'valid_loss', 'train_loss', 'train_acc', 'optimizer'}


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_generation_driver_end_to_end(tmp_path, capsys, precision):
    """finetune_generation() (main.py:214-321): args, data files, per-epoch train/valid/test, log + checkpoint; the FAD metrics that
    need the absent `shapesimilarity` package are reported as n/a / None, never as numbers."""
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    from pianobart_amd.finetune_generation import finetune_generation, get_args_generation
    a = get_args_generation([])
    assert (a.datasets, a.lr, a.epochs, a.batch_size, a.hs, a.layers, a.eval) == ('maestro', 2e-6, 500, 8, 1024, 8, False)
    root = str(tmp_path / 'data')
    os.makedirs(root)
    _write_finetune_sets(root, 'maestro', False, gen=True)
    vocab = os.path.join(os.path.dirname(GOLD), '..', 'pianobart_amd', 'data', 'octuple_vocab.json')
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        tr = finetune_generation(['--datasets', 'maestro', '--dataroot', root, '--dict_file', os.path.abspath(vocab), '--name', 'g', '--num_workers', '0',
                                  '--batch_size', '2', '--max_seq_len', str(S), '--hs', str(D), '--layers', '1', '--ffn_dims', '128', '--heads', '4',
                                  '--epochs', '2', '--nopretrain', '--lr', '1e-3', '--cuda_devices', '0', '--precision', precision])
        out = capsys.readouterr().out
        log = open('result/finetune/generation_g/log').read().split('\n')
        ck = torch.load('result/finetune/generation_g/model.ckpt', weights_only=False)
    finally:
        os.chdir(cwd)
    assert len(log) == 4 and log[1].startswith('Epoch 1: train_loss=') and 'train_fad=None' in log[1]
    assert 'FAD(BAR) Similarity: n/a' in out and 'FAD Similarity 0.0' not in out
    assert set(ck.keys()) == {'epoch', 'state_dict', 'valid_acc', 'valid_loss', 'train_loss', 'train_acc', 'optimizer'}
    assert any(k.startswith('mask_lm.proj.') for k in ck['state_dict']) and any(k.startswith('pianobart.') for k in ck['state_dict'])
    assert tr.test()[2:4] == (None, None)


@pytest.mark.gpu
def test_head_adamw_leaves_parameters_without_gradient_alone():
    """transformers.AdamW: `if p.grad is None: continue` -- no update, no decay, no step count; BART's dead `shared` table is not
    re-homed into the head optimizer's flat buffers at all."""
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    from oracle import pianobart_oracle as O
    from pianobart_amd.finetune import HeadAdamW
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n, generator=g).cuda()) for n in (10, 7, 33, 4)]
    dead = torch.nn.Parameter(torch.randn(5, generator=g).cuda())
    opt = HeadAdamW(ps + [dead], lr=1e-2, never=[dead])
    assert len(opt.params) == 4 and opt.P.numel() == 12 + 8 + 36 + 4
    ref = [p.detach().cpu().clone() for p in ps]
    m = [torch.zeros_like(r) for r in ref]; v = [torch.zeros_like(r) for r in ref]
    steps = [0, 0, 0, 0]
    for it in range(3):
        live = [0, 1, 2, 3] if it != 1 else [0, 2]                       # step 2: parameters 1 and 3 receive no gradient
        for i, p in enumerate(ps):
            p.grad = torch.randn(p.shape, generator=g).cuda() if i in live else None
        opt.step()
        for i in live:
            steps[i] += 1
            O.hf_adamw_step([ref[i]], [ps[i].grad.cpu()], [m[i]], [v[i]], step=steps[i], lr=1e-2)
    assert opt.steps == steps == [3, 2, 3, 2]
    for p, r in zip(ps, ref):
        assert _rel(p.detach(), r) < 1e-6
    assert dead.data_ptr() < opt.P.data_ptr() or dead.data_ptr() >= opt.P.data_ptr() + 4 * opt.P.numel()


@pytest.mark.gpu
def test_embeddings_and_mlm_forward_for_direct_callers():
    """Embeddings.forward (PianoBart.py:15-16) and MLM.forward (model.py:119-126) are fused away inside PianoBart / PianoBartLM but
    stay callable on their own, with gradients, and agree with the oracle; MLM on the backbone's hidden state reproduces the fused
    8-head GEMM of PianoBartLM.forward."""
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a HIP device')
    from oracle import pianobart_oracle as O
    from pianobart_amd import model as M
    m = M.PianoBartLM(M.PianoBart(M.BartConfig(**KW), E2W, W2E, precision='fp32'))
    randomize_params(m, 5)
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**KW), E2W, W2E))
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda().eval(); o = o.eval()
    ids = torch.randint(0, 134, (2, 7))
    e = m.pianobart.word_emb[1](ids.cuda())
    eo = o.pianobart.word_emb[1](ids)
    assert _rel(e.detach(), eo.detach()) < 1e-6
    e.sum().backward()
    eo.sum().backward()
    assert _rel(m.pianobart.word_emb[1].lut.weight.grad, o.pianobart.word_emb[1].lut.weight.grad) < 1e-6
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(2, S, seed=4)
    with torch.no_grad():
        fused = m(enc.cuda(), dec.cuda(), emask.cuda(), dmask.cuda())
        y = m.pianobart(enc.cuda(), dec.cuda(), emask.cuda(), dmask.cuda())
        sep = m.mask_lm(y)
        ref = o(enc, dec, emask, dmask)
    for a, b, r in zip(sep, fused, ref):
        assert a.shape == b.shape and _rel(a, b) < 1e-5 and _rel(a, r) < 1e-4
