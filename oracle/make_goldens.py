"""Generate tests/golden/*.npz by running the REAL reference (/root/reference + installed
transformers) in the build container, and cross-check oracle/pianobart_oracle.py against it.

Run:  python oracle/make_goldens.py          (only works where /root/reference exists)

Nothing here travels as a dependency: the GPU box and the test-suite only read the
committed .npz / .json vectors. TEST INFRASTRUCTURE ONLY.
"""
import hashlib
import io
import json
import os
import pickle
import random
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from transformers import BartConfig as HFBartConfig, BartModel  # noqa: F401  (import first)
sys.modules['transformers'].AdamW = torch.optim.AdamW               # removed upstream; pretrain.py:3
sys.modules['shapesimilarity'] = types.ModuleType('shapesimilarity')
sys.modules['shapesimilarity'].shape_similarity = lambda *a, **k: 0.0

import PianoBart as ref_pb            # noqa: E402  reference
import model as ref_model             # noqa: E402  reference
import pretrain as ref_pretrain       # noqa: E402  reference

from oracle import pianobart_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)

with open(os.path.join(REF, 'Data', 'Octuple.pkl'), 'rb') as f:
    E2W, W2E = pickle.load(f)


def sd_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def cfg_pair(S, d, L, f, h, dropout=0.1):
    kw = dict(max_position_embeddings=S, d_model=d, encoder_layers=L, decoder_layers=L,
              encoder_ffn_dim=f, decoder_ffn_dim=f, encoder_attention_heads=h,
              decoder_attention_heads=h, dropout=dropout)
    return HFBartConfig(**kw), O.BartConfig(**kw)


def build_pair(S, d, L, f, h, seed, dropout=0.1, lm=True):
    """Oracle model with seeded, *non-trivial* weights; same weights loaded into the reference
    with strict=True (this also pins the state_dict key layout, SURVEY b-3)."""
    hf_cfg, o_cfg = cfg_pair(S, d, L, f, h, dropout)
    o = O.PianoBart(o_cfg, E2W, W2E)
    r = ref_pb.PianoBart(hf_cfg, E2W, W2E)
    if lm:
        o = O.PianoBartLM(o)
        r = ref_model.PianoBartLM(r)
    O_randomize(o, seed)
    assert list(o.state_dict().keys()) == list(r.state_dict().keys()), "state_dict key order differs"
    for (k1, v1), (k2, v2) in zip(o.state_dict().items(), r.state_dict().items()):
        assert v1.shape == v2.shape, (k1, v1.shape, v2.shape)
    r.load_state_dict(o.state_dict(), strict=True)
    return o, r


def O_randomize(model, seed):
    from tests.golden_util import randomize_params
    randomize_params(model, seed)


def synth_batch(B, S, seed, min_len=None):
    from tests.golden_util import synth_octuple_batch
    return synth_octuple_batch(B, S, seed, min_len)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def save(name, **arrs):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def g_vocab():
    out = os.path.join(ROOT, 'pianobart_amd', 'data')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'octuple_vocab.json'), 'w') as f:
        json.dump({'e2w': E2W}, f)
    print('vocab sha', hashlib.sha256(open(os.path.join(REF, 'Data', 'Octuple.pkl'), 'rb').read()).hexdigest()[:16])


def g1_forward():
    """G1/G2/G3: cfg-1 shape forward (eval), encoder-only branch, loss/acc/argmax."""
    o, r = build_pair(128, 128, 2, 512, 4, seed=11)
    o.eval(); r.eval()
    enc, dec, loss_mask, emask, dmask, target = synth_batch(2, 128, seed=5)
    # make one decoder row "padded then non-padded" 
    dmask = dmask.clone(); dmask[1, 3] = 0
    with torch.no_grad():
        yo = o(enc, dec, emask, dmask)
        yr = r(enc, dec, emask, dmask)
        ho = o.pianobart(enc, dec, emask, dmask)
        hr = r.pianobart(enc, dec, emask, dmask)
        eo = o.pianobart(enc, None, emask, None)
        er = r.pianobart(enc, None, emask, None)
    for i in range(8):
        assert rel(yo[i], yr[i]) < 2e-5, ('logits', i, rel(yo[i], yr[i]))
    assert rel(ho.last_hidden_state, hr.last_hidden_state) < 2e-5
    assert rel(ho.encoder_last_hidden_state, hr.encoder_last_hidden_state) < 2e-5
    assert rel(eo.last_hidden_state, er.last_hidden_state) < 2e-5
    tot_r, losses_r, accs_r, arg_r = O.pretrain_loss([t.clone() for t in yr], target, loss_mask, E2W)
    # loss through the reference's own compute_loss (pretrain.py:112-118)
    tr = ref_pretrain.Pretrainer.__new__(ref_pretrain.Pretrainer)
    tr.loss_func = torch.nn.CrossEntropyLoss(reduction='none')
    ref_losses = [tr.compute_loss(yr[i].permute(0, 2, 1), target[..., i], loss_mask[..., i]) for i in range(8)]
    n_tok = [len(E2W[e]) for e in E2W]
    ref_total = sum(l * w for l, w in zip(ref_losses, n_tok)) / sum(n_tok)
    assert abs(float(ref_total) - float(tot_r)) < 1e-6
    save('g1_forward_cfg1.npz', enc=enc.to(torch.int16), dec=dec.to(torch.int16), emask=emask, dmask=dmask,
         loss_mask=loss_mask.to(torch.uint8), target=target.to(torch.int16),
         logits=torch.cat(yr, dim=-1), hidden=hr.last_hidden_state,
         enc_hidden=hr.encoder_last_hidden_state, enc_only_hidden=er.last_hidden_state,
         total_loss=ref_total, head_losses=torch.stack(ref_losses), head_acc=torch.stack(accs_r),
         argmax=arg_r.to(torch.int16), sd_sha=np.frombuffer(sd_checksum(r.state_dict()).encode(), dtype=np.uint8))


def g4_grads():
    """G4/G5: dropout=0 train step on the reference: grads of named tensors, global norm,
    then one HF-AdamW step (restated formula; pinned transformers.AdamW absent) checksums."""
    o, r = build_pair(64, 64, 2, 128, 4, seed=23, dropout=0.0)
    o.train(); r.train()
    enc, dec, loss_mask, emask, dmask, target = synth_batch(2, 64, seed=9)
    names = ['pianobart.word_emb.0.lut.weight', 'pianobart.word_emb.5.lut.weight',
             'pianobart.encoder_linear.weight', 'pianobart.encoder_linear.bias',
             'pianobart.bart.encoder.embed_positions.weight',
             'pianobart.bart.encoder.layernorm_embedding.weight',
             'pianobart.bart.encoder.layers.0.self_attn.q_proj.weight',
             'pianobart.bart.encoder.layers.1.fc1.bias',
             'pianobart.bart.decoder.layers.0.encoder_attn.k_proj.weight',
             'pianobart.bart.decoder.layers.1.encoder_attn.v_proj.bias',
             'pianobart.bart.decoder.layers.1.self_attn.out_proj.weight',
             'pianobart.bart.decoder.layers.1.final_layer_norm.bias',
             'pianobart.bart.decoder.layers.0.fc2.weight',
             'mask_lm.proj.2.weight', 'mask_lm.proj.7.bias']
    outs = {}
    for tag, m in (('o', o), ('r', r)):
        m.zero_grad()
        y = m(enc, dec, emask, dmask)
        total, losses, accs, arg = O.pretrain_loss(y, target, loss_mask, E2W)
        total.backward()
        outs[tag] = (total.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert abs(float(outs['o'][0]) - float(outs['r'][0])) < 1e-5
    for k in names:
        assert rel(outs['o'][1][k], outs['r'][1][k]) < 5e-4, (k, rel(outs['o'][1][k], outs['r'][1][k]))
    gr = outs['r'][1]
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in gr.values())).float()
    # per-parameter grad norms for every parameter (cheap, pins all of backward)
    all_names = sorted(gr.keys())
    per_norm = torch.stack([gr[k].double().norm().float() for k in all_names])
    # clip + HF AdamW step on the reference's parameters
    params = [p for k, p in r.named_parameters() if k in gr]
    pnames = [k for k, p in r.named_parameters() if k in gr]
    grads = [gr[k].clone() for k in pnames]
    O.clip_grad_norm(grads, 3.0)
    m1 = [torch.zeros_like(p) for p in params]
    v1 = [torch.zeros_like(p) for p in params]
    with torch.no_grad():
        newp = [p.detach().clone() for p in params]
        O.hf_adamw_step(newp, grads, m1, v1, step=1, lr=2e-5)
        delta = torch.stack([(n - p).double().norm().float() for n, p in zip(newp, params)])
        psum = torch.stack([n.double().sum().float() for n in newp])
    save('g4_grads_small.npz', enc=enc.to(torch.int16), dec=dec.to(torch.int16), emask=emask, dmask=dmask,
         loss_mask=loss_mask.to(torch.uint8), target=target.to(torch.int16), total_loss=outs['r'][0],
         grad_norm=gnorm, per_param_grad_norm=per_norm,
         param_names=np.array(all_names), step_param_names=np.array(pnames),
         adamw_delta_norm=delta, adamw_param_sum=psum,
         **{'grad__' + k: gr[k] for k in names})


def g6_gen_mask():
    """G6: exact gen_mask outputs of the reference under fixed seeds (seeds 0-4 x choices 1-5)."""
    o, r = build_pair(64, 32, 1, 64, 4, seed=3, lm=False)
    tr = ref_pretrain.Pretrainer.__new__(ref_pretrain.Pretrainer)
    tr.pianobart = r
    tr.max_seq_len = 64
    tr.mask_percent = 0.15
    tr.Lseq = list(range(64))
    tr.Lseq_element = list(range(64 * 8))
    tr.device = torch.device('cpu')
    corr = O.Corruptor(o, 64, 0.15)
    ids = synth_batch(1, 64, seed=17, min_len=64)[5][0].long()
    out = {'ids': ids.to(torch.int16)}
    for choice in range(1, 6):
        for seed in range(5):
            random.seed(seed); np.random.seed(seed)
            mr, pr = tr.gen_mask(ids.clone(), choice)
            random.seed(seed); np.random.seed(seed)
            mo, po = corr.gen_mask(ids.clone(), choice)
            mr_t = torch.as_tensor(np.asarray(mr)); pr_t = torch.as_tensor(np.asarray(pr))
            assert mr_t.shape == mo.shape and bool((mr_t.double() == mo.double()).all()), (choice, seed)
            assert bool((pr_t.double() == torch.as_tensor(np.asarray(po)).double()).all()), (choice, seed)
            assert mr_t.dtype == mo.dtype, (choice, seed, mr_t.dtype, mo.dtype)
            out['masked_c%d_s%d' % (choice, seed)] = mr_t.to(torch.int16)
            out['pos_c%d_s%d' % (choice, seed)] = pr_t.to(torch.uint8)
    # whole-batch construction (pretrain.py:127-153) under one seed
    random.seed(7); np.random.seed(7)
    batch = synth_batch(3, 64, seed=21)[5]
    enc_o, dec_o, lm_o, em_o, dm_o = O.pretrain_batch(corr, batch)
    out.update(batch=batch.to(torch.int16), batch_enc=enc_o.to(torch.int16), batch_dec=dec_o.to(torch.int16),
               batch_loss_mask=lm_o.to(torch.uint8), batch_emask=em_o, batch_dmask=dm_o)
    save('g6_gen_mask.npz', **out)


def g7_sampling():
    """G7: nucleus/sampling known answers and RNG stream position."""
    rng = np.random.default_rng(4)
    logits = torch.from_numpy(rng.normal(size=(6, 1, 40)).astype(np.float32) * 3)
    res = {}
    picks_ref, picks_o = [], []
    np.random.seed(2023)
    for i in range(6):
        for p, t in ((1, 1.2), (0.9, 1.0), (0.9, 2.0)):
            picks_ref.append(ref_model.sampling(logits[i], p, t))
    after_ref = np.random.rand()
    np.random.seed(2023)
    for i in range(6):
        for p, t in ((1, 1.2), (0.9, 1.0), (0.9, 2.0)):
            picks_o.append(O.sampling(logits[i], p, t))
    after_o = np.random.rand()
    assert picks_ref == picks_o and after_ref == after_o
    save('g7_sampling.npz', logits=logits, picks=np.array(picks_ref), rng_after=after_ref)


def g8_generate():
    """G8: generate trace (token ids) on a tiny model under np.random.seed(2023)."""
    o, r = build_pair(24, 64, 2, 128, 4, seed=31)
    o.eval(); r.eval()
    enc, dec, loss_mask, emask, dmask, target = synth_batch(1, 24, seed=13, min_len=20)
    with torch.no_grad():
        np.random.seed(2023)
        gr = r(enc, None, emask, None, generate=True)
        np.random.seed(2023)
        go = o(enc, None, emask, None, generate=True)
        # pre-sampling logits at step 0 (decoder = SOS + PAD rows)
        pad = torch.from_numpy(r.pianobart.pad_word_np)
        d0 = pad.repeat(1, 24, 1); d0[:, 0, :] = torch.tensor(r.pianobart.sos_word_np)
        dm0 = torch.zeros_like(emask); dm0[:, 0] = 1
        step0 = torch.cat(r(enc, d0, emask, dm0), dim=-1)[:, 0]
    assert bool((gr == go).all()), 'generate trace differs'
    save('g8_generate.npz', enc=enc.to(torch.int16), emask=emask, tokens=gr.to(torch.int16), step0_logits=step0,
         sd_sha=np.frombuffer(sd_checksum(r.state_dict()).encode(), dtype=np.uint8))


def g9_state_dict():
    """G9: key list + shapes for (2L,128) and (12L,768)."""
    out = {}
    for tag, (S, d, L, f, h) in {'cfg1': (128, 128, 2, 512, 4), 'cfg2': (1024, 768, 12, 3072, 12)}.items():
        hf_cfg, _ = cfg_pair(S, d, L, f, h)
        with torch.device('meta'):
            r = ref_model.PianoBartLM(ref_pb.PianoBart(hf_cfg, E2W, W2E))
        out[tag] = [[k, list(v.shape)] for k, v in r.state_dict().items()]
        out[tag + '_n_params'] = sum(p.numel() for p in r.parameters())
    with open(os.path.join(GOLD, 'g9_state_dict.json'), 'w') as f:
        json.dump(out, f)
    print('wrote g9_state_dict.json', out['cfg1_n_params'], out['cfg2_n_params'])


def g10_cfg2_spot():
    """G10: cfg-2 shape spot check, B=1 S=1024 eval forward on the reference."""
    o, r = build_pair(1024, 768, 12, 3072, 12, seed=41)
    r.eval()
    enc, dec, loss_mask, emask, dmask, target = synth_batch(1, 1024, seed=19)
    with torch.no_grad():
        y = torch.cat(r(enc, dec, emask, dmask), dim=-1)[0]      # (1024,1280)
    offs = np.cumsum([0] + r.pianobart.n_tokens)
    arg = torch.stack([y[:, offs[i]:offs[i + 1]].argmax(-1) for i in range(8)], dim=-1)
    # top-2 gap per head/position, so the test can skip near-ties
    gap = torch.stack([(lambda t: t[:, 0] - t[:, 1])(y[:, offs[i]:offs[i + 1]].topk(2, dim=-1).values) for i in range(8)], dim=-1)
    rows = np.linspace(0, 1023, 64).astype(np.int64)
    save('g10_cfg2_spot.npz', argmax=arg.to(torch.int16), top2_gap=gap, rows=rows, logit_rows=y[rows],
         logit_absmax=y.abs().max(), sd_sha=np.frombuffer(sd_checksum(r.state_dict()).encode(), dtype=np.uint8))


if __name__ == '__main__':
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['vocab', 'g1', 'g4', 'g6', 'g7', 'g8', 'g9', 'g10']
    fns = dict(vocab=g_vocab, g1=g1_forward, g4=g4_grads, g6=g6_gen_mask, g7=g7_sampling, g8=g8_generate,
               g9=g9_state_dict, g10=g10_cfg2_spot)
    for w in which:
        if w in ('g11', 'g12', 'g13', 'g14'):
            continue
        print('==', w)
        fns[w]()


def g12_finetune_heads():
    """G12: fine-tune heads: SequenceClassification (composer-like, 8 classes) and TokenClassification in both forms
    (4 classes: ordinary decoder input; 8 classes: the velocity task's decoder label-embedding swap), dropout off: logits, loss,
    gradient norm and a few named gradients from the REAL reference; the oracle restatement is asserted against it first."""
    S, d, L, f, h = 64, 128, 2, 256, 4
    hf_cfg, o_cfg = cfg_pair(S, d, L, f, h, dropout=0.0)
    enc, dec, loss_mask, emask, dmask, target = synth_batch(2, S, seed=21)
    out = {}
    for tag, mk_o, mk_r in (
            ('seq', lambda pb: O.SequenceClassification(pb, 8, d), lambda pb: ref_model.SequenceClassification(pb, 8, d)),
            ('tok4', lambda pb: O.TokenClassification(pb, 4, d), lambda pb: ref_model.TokenClassification(pb, 4, d)),
            ('tok8', lambda pb: O.TokenClassification(pb, 8, d), lambda pb: ref_model.TokenClassification(pb, 8, d))):
        o = mk_o(O.PianoBart(o_cfg, E2W, W2E)); r = mk_r(ref_pb.PianoBart(hf_cfg, E2W, W2E))
        O_randomize(o, 31)
        assert list(o.state_dict().keys()) == list(r.state_dict().keys()), (tag, 'state_dict keys differ')
        r.load_state_dict(o.state_dict(), strict=True)
        for mdl in (o, r):
            mdl.train()
            for m in mdl.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
        g = torch.Generator().manual_seed(7)
        if tag == 'seq':
            y = torch.randint(0, 8, (2,), generator=g)
            run = lambda mdl: mdl(input_ids_encoder=enc, encoder_attention_mask=emask)
            lossf = lambda yh: O.finetune_loss(yh, y, None, True)
        elif tag == 'tok4':
            y = torch.randint(0, 4, (2, S), generator=g)
            run = lambda mdl: mdl(input_ids_encoder=enc, input_ids_decoder=enc, encoder_attention_mask=emask, decoder_attention_mask=emask)
            lossf = lambda yh: O.finetune_loss(yh, y, emask, False)
        else:
            y = torch.randint(0, 7, (2, S), generator=g)
            y_shift = torch.zeros_like(y) + 7
            y_shift[:, 1:] = y[:, :-1]
            attn_shift = torch.zeros_like(emask); attn_shift[:, 1:] = emask[:, :-1]; attn_shift[:, 0] = emask[:, 0]
            run = lambda mdl: mdl(input_ids_encoder=enc, input_ids_decoder=y_shift, encoder_attention_mask=emask, decoder_attention_mask=attn_shift)
            lossf = lambda yh: O.finetune_loss(yh, y, emask, False)
            out['tok8_y_shift'] = y_shift; out['tok8_attn_shift'] = attn_shift
        res = {}
        for name, mdl in (('o', o), ('r', r)):
            mdl.zero_grad()
            yh = run(mdl)
            loss = lossf(yh)
            loss.backward()
            res[name] = (yh.detach(), loss.detach(), {k: p.grad.detach().clone() for k, p in mdl.named_parameters() if p.grad is not None})
        assert rel(res['o'][0], res['r'][0]) < 2e-5, (tag, rel(res['o'][0], res['r'][0]))
        assert abs(float(res['o'][1]) - float(res['r'][1])) < 1e-5 * abs(float(res['r'][1]))
        gr = res['r'][2]
        gmax = max(float(v.abs().max()) for v in gr.values())
        for k in gr:       # k_proj.bias gradients are mathematically zero (softmax shift invariance): noise only
            assert rel(res['o'][2][k], gr[k]) < 5e-4 or float(gr[k].abs().max()) < 1e-5 * gmax, (tag, k, rel(res['o'][2][k], gr[k]))
        gnorm = torch.sqrt(sum((v.double() ** 2).sum() for v in gr.values())).float()
        names = [k for k in gr if k.startswith('classifier') or k.startswith('attention') or 'decoder_emb' in k or 'decoder_linear' in k]
        names += ['pianobart.bart.decoder.layers.0.self_attn.q_proj.weight', 'pianobart.word_emb.3.lut.weight']
        out[tag + '_logits'] = res['r'][0]; out[tag + '_loss'] = res['r'][1]; out[tag + '_gnorm'] = gnorm; out[tag + '_y'] = y
        out[tag + '_sd'] = np.array(sd_checksum(o.state_dict()))
        out[tag + '_grad_names'] = np.array(names)
        for i, k in enumerate(names):
            out['%s_grad_%d' % (tag, i)] = gr[k]
        print('G12', tag, 'logits rel', rel(res['o'][0], res['r'][0]), 'loss', float(res['r'][1]), 'gnorm', float(gnorm))
    out['enc'] = enc.to(torch.int16); out['emask'] = emask
    save('g12_finetune_heads.npz', **out)


if __name__ == '__main__' and 'g12' in sys.argv[1:]:
    g12_finetune_heads()


def g11_pretrain_artifacts():
    """G11: run the reference's own main.pretrain() (cfg 1: 2L/128d, S=128, B=2, 10 synthetic sequences, 1 epoch, CPU) in a
    temp cwd and record the artefacts a drop-in harness must reproduce: log line, stdout line formats, checkpoint keys."""
    import contextlib, io, re, tempfile
    from tests.golden_util import synth_octuple_batch
    import main as ref_main            # reference
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, 'Data', 'output_pretrain', 'syn')
    os.makedirs(root)
    seqs = synth_octuple_batch(10, 128, seed=77)[5].numpy().astype(np.int64)
    for name, part in (('train', seqs[:6]), ('test', seqs[6:8]), ('valid', seqs[8:])):
        np.save(os.path.join(root, 'syn_%s_split.npy' % name), part)
    cwd, argv = os.getcwd(), sys.argv
    os.chdir(tmp)
    sys.argv = ['main.py', '--dict_file', os.path.join(REF, 'Data', 'Octuple.pkl'), '--name', 't', '--datasets', 'syn', '--num_workers', '0',
                '--batch_size', '2', '--max_seq_len', '128', '--hs', '128', '--layers', '2', '--ffn_dims', '512', '--heads', '4',
                '--epochs', '1', '--cpu', '--cuda_devices', '0']
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            ref_main.pretrain()
        log = open('result/pretrain/t/log').read()
        ck = torch.load('result/pretrain/t/model.ckpt', weights_only=False)
    finally:
        os.chdir(cwd); sys.argv = argv
    out = buf.getvalue()
    rec = {'log': log, 'ckpt_keys': sorted(ck.keys()), 'n_state_dict': len(ck['state_dict']), 'state_dict_keys': list(ck['state_dict'].keys()),
           'files': sorted(os.listdir(os.path.join(tmp, 'result', 'pretrain', 't'))),
           'stdout_loss_line': [l for l in out.splitlines() if l.startswith('Loss: ')][0],
           'stdout_acc_line': [l for l in out.splitlines() if l.startswith('Acc: ')][0],
           'stdout_epoch_line': [l for l in out.splitlines() if l.startswith('epoch: ')][0]}
    with open(os.path.join(GOLD, 'g11_pretrain_artifacts.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    print('wrote g11_pretrain_artifacts.json'); print(rec['log'][:200]); print(rec['stdout_loss_line']); print(rec['ckpt_keys'], rec['files'])


if __name__ == '__main__' and 'g11' in sys.argv[1:]:
    g11_pretrain_artifacts()


def g13_octuple_midi():
    """G13: MIDI_to_encoding / encoding_to_MIDI / padding of the REAL reference (Data/data_generation/convert.py) on
    a synthetic song. The reference imports the third-party `miditoolkit` (absent here) only for its plain data containers and the
    file parser; the converters themselves touch nothing but attributes, so the containers are stood in for by attribute bags
    (data holders, no logic) and the file parser is not used. The vectors: the song (notes, time-signature and tempo changes), the
    reference's encoding of it, its padded forms, and the reference's decoding of that encoding back into notes / changes."""
    class Bag:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    mt = types.ModuleType('miditoolkit')
    mt.midi = types.ModuleType('miditoolkit.midi'); mt.midi.parser = types.ModuleType('miditoolkit.midi.parser')
    mt.containers = types.ModuleType('miditoolkit.containers')
    mt.midi.parser.MidiFile = lambda *a, **k: Bag(ticks_per_beat=480, instruments=[], time_signature_changes=[], tempo_changes=[])
    mt.containers.Instrument = lambda program, is_drum, name: Bag(program=program, is_drum=is_drum, name=name, notes=[])
    mt.containers.Note = lambda start, end, pitch, velocity: Bag(start=start, end=end, pitch=pitch, velocity=velocity)
    mt.containers.TimeSignature = lambda numerator, denominator, time: Bag(numerator=numerator, denominator=denominator, time=time)
    mt.containers.TempoChange = lambda tempo, time: Bag(tempo=tempo, time=time)
    for name, mod in (('miditoolkit', mt), ('miditoolkit.midi', mt.midi), ('miditoolkit.midi.parser', mt.midi.parser), ('miditoolkit.containers', mt.containers)):
        sys.modules[name] = mod
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_convert', os.path.join(REF, 'Data', 'data_generation', 'convert.py'))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    rng = np.random.default_rng(13)
    tpb = 384
    insts = [Bag(program=0, is_drum=False, name='PIANO', notes=[]), Bag(program=40, is_drum=False, name='x', notes=[]), Bag(program=0, is_drum=True, name='d', notes=[])]
    t = 0
    for _ in range(400):
        t += int(rng.integers(0, tpb))
        k = int(rng.choice(3, p=[0.6, 0.3, 0.1]))
        insts[k].notes.append(Bag(start=t, end=t + int(rng.integers(1, 6 * tpb)), pitch=int(rng.integers(21, 108)), velocity=int(rng.integers(1, 128))))
    song = Bag(ticks_per_beat=tpb, instruments=insts,
               time_signature_changes=[Bag(numerator=4, denominator=4, time=0), Bag(numerator=3, denominator=4, time=16 * tpb), Bag(numerator=6, denominator=8, time=40 * tpb),
                                       Bag(numerator=12, denominator=4, time=70 * tpb)],
               tempo_changes=[Bag(tempo=96.0, time=0), Bag(tempo=133.7, time=30 * tpb), Bag(tempo=300.0, time=60 * tpb), Bag(tempo=10.0, time=80 * tpb)])
    enc = conv.MIDI_to_encoding(song, task='pretrain')
    notes = np.array([[n.start, n.end, n.pitch, n.velocity, i.program, int(i.is_drum)] for i in insts for n in i.notes], dtype=np.int64)
    melodic = [e for e in enc if e[2] <= 128 and e[3] <= 255]                 # rows encoding_to_MIDI can place (the reference codes drums past the vocabulary)
    back = conv.encoding_to_MIDI(melodic)
    back_notes = np.array(sorted([n.start, n.end, n.pitch, n.velocity, i.program, int(i.is_drum)] for i in back.instruments for n in i.notes), dtype=np.int64)
    save('g13_octuple_midi.npz', ticks_per_beat=tpb, notes=notes,
         ts=np.array([[c.time, c.numerator, c.denominator] for c in song.time_signature_changes], dtype=np.int64),
         tp=np.array([[c.time, c.tempo] for c in song.tempo_changes], dtype=np.float64),
         encoding=np.array(enc, dtype=np.int64),
         padded_1024=np.array(conv.padding('x', list(enc[:300]), 1024), dtype=np.int64),
         padded_cut_head=np.array(conv.padding('x', list(enc), 256, last=False), dtype=np.int64),
         padded_cut_tail=np.array(conv.padding('x', list(enc), 256, last=True), dtype=np.int64),
         melodic=np.array(melodic, dtype=np.int64), back_notes=back_notes,
         back_ts=np.array([[c.time, c.numerator, c.denominator] for c in back.time_signature_changes], dtype=np.int64),
         back_tp=np.array([[c.time, c.tempo] for c in back.tempo_changes], dtype=np.float64),
         tables=np.array([len(conv.ts_list), len(conv.dur_enc), len(conv.dur_dec), conv.t2e((6, 8)), conv.d2e(1000), conv.e2d(77), conv.b2e(133.7)], dtype=np.int64))
    print('g13: %d notes -> %d rows (%d placeable), decoded %d notes, %d ts / %d tempo changes' % (len(notes), len(enc), len(melodic), len(back_notes),
                                                                                             len(back.time_signature_changes), len(back.tempo_changes)))


if __name__ == '__main__' and 'g13' in sys.argv[1:]:
    g13_octuple_midi()


def g14_generation_trainer():
    """G14: the REAL reference `finetune_generation.GenerationTrainer` (finetune_generation.py:57-272) on a seeded tiny
    model, CPU: one `iteration` in test mode (mode 2: loss, accuracies, the argmax ids `all_output`), one in valid mode and one in
    train mode on a single batch (the per-head CE values the reference's `compute_loss` returns, the value `clip_grad_norm_` reports
    = the gradient norm before clipping, and a few named gradients). Only quantities that do not depend on the optimizer flavour are
    stored (`transformers.AdamW` 4.29.2 is absent here, SURVEY a-11): the train-mode loader holds ONE batch, so nothing is computed
    after the update. `y_shift = x` (finetune_generation.py:155) and the head weights 0.3 / 1.5 / 1 (:239-250) are exercised as the
    reference runs them, not as a formula restated in the test. The oracle forward is asserted against the same run first."""
    import contextlib
    import finetune_generation as ref_fg          # reference (needs the `shapesimilarity` stand-in registered at the top: FAD is a host metric)
    S, d, L, f, h = 64, 64, 1, 128, 4
    hf_cfg, o_cfg = cfg_pair(S, d, L, f, h, dropout=0.0)
    x = synth_batch(4, S, seed=31)[5]
    y = synth_batch(4, S, seed=32)[5]
    o = O.PianoBartLM(O.PianoBart(o_cfg, E2W, W2E))
    O_randomize(o, 13)
    rpb = ref_pb.PianoBart(hf_cfg, E2W, W2E)
    rlm = ref_model.PianoBartLM(rpb)
    rlm.load_state_dict(o.state_dict(), strict=True)
    tr = ref_fg.GenerationTrainer(rpb, [(x, y)], [(x, y)], [(x, y)], 1e-3, (4, S, 8), True, [0], model=rlm)
    ce_log = []
    orig_cl = tr.compute_loss
    tr.compute_loss = lambda p, t, m: (lambda v: (ce_log.append(float(v)), v)[1])(orig_cl(p, t, m))
    out = {'x': x.to(torch.int16), 'y': y.to(torch.int16), 'sd': np.array(sd_checksum(o.state_dict()))}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        loss, accs, fb, fa, all_output = tr.test()
    out['test_loss'], out['test_accs'], out['test_all_output'] = loss, np.array(accs), all_output.to(torch.int16)
    out['test_head_ce'] = np.array(ce_log[-8:], dtype=np.float64)
    out['test_stdout'] = np.array([l for l in buf.getvalue().splitlines() if l.startswith(('Loss:', 'Acc:'))])
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        vloss, vaccs, _, _ = tr.valid()
    assert vloss == loss and vaccs == accs
    # the oracle on the same inputs (y_shift = x, masks from the bar column)
    mask = (x[:, :, 0] != 256).float()
    with torch.no_grad():
        yh = o.eval()(x, x, mask, mask)
    assert torch.equal(torch.stack([t.argmax(-1) for t in yh], -1), all_output.long()), 'oracle argmax differs from the reference trainer'
    for i in range(8):
        ce = torch.nn.functional.cross_entropy(yh[i].permute(0, 2, 1), y[..., i], reduction='none')
        assert abs(float((ce * mask).sum() / mask.sum()) - ce_log[i]) < 2e-5 * ce_log[i], (i, ce_log[i])
    # train mode, ONE batch: values before the update
    grads = {}

    def clip_spy(params, max_norm):
        params = list(params)
        for (k, _), p in zip(tr.model.named_parameters(), params):
            if p.grad is not None:
                grads[k] = p.grad.detach().clone()
        return torch.nn.utils.clip_grad_norm_(params, max_norm)
    saved = ref_fg.clip_grad_norm_
    ref_fg.clip_grad_norm_ = clip_spy
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            tloss, taccs, _, _ = tr.train()
    finally:
        ref_fg.clip_grad_norm_ = saved
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    names = ['mask_lm.proj.3.weight', 'mask_lm.proj.0.bias', 'pianobart.bart.decoder.layers.0.fc1.weight', 'pianobart.bart.encoder.layers.0.self_attn.q_proj.weight',
             'pianobart.word_emb.3.lut.weight', 'pianobart.encoder_linear.weight', 'pianobart.bart.decoder.layernorm_embedding.weight', 'pianobart.bart.decoder.embed_positions.weight']
    out['train_loss'], out['train_accs'], out['train_head_ce'] = tloss, np.array(taccs), np.array(ce_log[-8:], dtype=np.float64)
    out['train_gnorm'] = float(gnorm)
    out['train_grad_names'] = np.array(names)
    for i, k in enumerate(names):
        out['train_grad_%d' % i] = grads[k]
    out['train_stdout'] = np.array([l for l in buf.getvalue().splitlines() if l.startswith(('Loss:', 'Acc:'))])
    assert tloss == loss, (tloss, loss)                       # dropout 0: train-mode forward = eval-mode forward
    print('G14 test loss', loss, 'accs', accs, 'gnorm', float(gnorm), 'head ce', [round(v, 5) for v in ce_log[:8]])
    print(out['test_stdout'][0]); print(out['train_stdout'][0])
    save('g14_generation_trainer.npz', **out)


if __name__ == '__main__' and 'g14' in sys.argv[1:]:
    g14_generation_trainer()
