"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded
inputs and against the committed golden vectors captured from the reference (tests/golden/*.npz).

Tolerances (BASELINE.json north_star / SURVEY 8d): fp32 instantiation: logits rel = max|a-b|/max|b|
<= 1e-3 (we assert 1e-4), argmax ids identical wherever the reference top-2 gap > 1e-4*max|b|, loss
rel <= 1e-4. The bf16 instantiation is reported against a looser, explicit bound (it cannot meet
1e-3 through 24 post-LN layers; SURVEY 7 hard part 1).
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLD, load_vocab, randomize_params, sd_checksum, synth_octuple_batch

pytestmark = pytest.mark.gpu
E2W, W2E = load_vocab()


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _cfg(S, d, L, f, h, dropout=0.1):
    from pianobart_amd.model import BartConfig
    return BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=L, decoder_layers=L, encoder_ffn_dim=f,
                      decoder_ffn_dim=f, encoder_attention_heads=h, decoder_attention_heads=h, dropout=dropout)


def _lm(S, d, L, f, h, seed, precision, dropout=0.1):
    from pianobart_amd.model import PianoBart, PianoBartLM
    m = PianoBartLM(PianoBart(_cfg(S, d, L, f, h, dropout), E2W, W2E, precision=precision))
    randomize_params(m, seed)
    return m


# bounds of the bf16 throughput instantiation in the shape sweep (measured maxima x ~2, see the test's printed lines)
BF16_LOGITS, BF16_LOSS, BF16_NORM = 5e-2, 2e-3, 2e-2          # measured over the 8 shapes (round 6): 2.6e-2, 5.4e-4, 6.6e-3


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _sha(z):
    return bytes(z['sd_sha']).decode()


def test_state_dict_layout_matches_reference():
    g = json.load(open(os.path.join(GOLD, 'g9_state_dict.json')))
    m = _lm(128, 128, 2, 512, 4, 0, 'fp32')
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g['cfg1']


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-4), ('bf16', 2.5e-2)])        # measured: 1.2e-6, 1.9e-5, 1.04e-2
def test_g1_forward_golden(precision, tol):
    _need_gpu()
    z = np.load(os.path.join(GOLD, 'g1_forward_cfg1.npz'))
    m = _lm(128, 128, 2, 512, 4, 11, precision).eval()
    assert sd_checksum(m.state_dict()) == _sha(z)          # identical weights to the reference run
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(2, 128, seed=5)]
    dmask = torch.from_numpy(z['dmask']).cuda()
    with torch.no_grad():
        y = m(enc, dec, emask, dmask)
        logits = torch.cat(y, dim=-1)
        h = m.pianobart(enc, dec, emask, dmask)
        e = m.pianobart(enc, None, emask, None)
    gl = torch.from_numpy(z['logits'])
    assert [tuple(t.shape) for t in y] == [(2, 128, n) for n in [262, 134, 135, 262, 134, 38, 260, 55]]
    r = _rel(logits, gl)
    print('logits rel (%s) = %.3e' % (precision, r))
    assert r < tol
    assert _rel(h.last_hidden_state, torch.from_numpy(z['hidden'])) < tol
    assert _rel(h.encoder_last_hidden_state, torch.from_numpy(z['enc_hidden'])) < tol
    assert _rel(e.last_hidden_state, torch.from_numpy(z['enc_only_hidden'])) < tol
    # argmax: identical wherever the reference's top-2 gap is not a near-tie
    offs = np.cumsum([0, 262, 134, 135, 262, 134, 38, 260, 55])
    for i in range(8):
        seg = gl[..., offs[i]:offs[i + 1]]
        top2 = seg.topk(2, dim=-1).values
        clear = (top2[..., 0] - top2[..., 1]) > {'fp32': 1e-4, 'bf16x3': 1e-4, 'bf16': 5e-2}[precision] * float(gl.abs().max())
        mine = logits[..., offs[i]:offs[i + 1]].argmax(-1).cpu()
        assert bool((mine[clear] == torch.from_numpy(z['argmax'][..., i].astype(np.int64))[clear]).all()), 'head %d argmax' % i
        if precision != 'bf16':
            assert float(clear.float().mean()) > 0.99


@pytest.mark.parametrize('precision,ltol', [('fp32', 1e-4), ('bf16x3', 1e-4), ('bf16', 2e-2)])
def test_g1_fused_loss_acc_argmax(precision, ltol):
    """The fused K9 kernel (no D2H logits) reproduces pretrain.py:163-189 on the golden batch."""
    _need_gpu()
    from pianobart_amd import ops
    z = np.load(os.path.join(GOLD, 'g1_forward_cfg1.npz'))
    m = _lm(128, 128, 2, 512, 4, 11, precision).eval().cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(2, 128, seed=5)]
    dmask = torch.from_numpy(z['dmask']).cuda()
    eng = m._get_engine()
    eng.bind(enc.device)
    sums = eng.loss_and_grads(ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(),
                              emask, dmask, train=False).cpu().double()
    head_loss = sums[0:8] / sums[8:16]
    head_acc = sums[16:24] / sums[8:16]
    w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
    total = float((head_loss * w).sum() / w.sum())
    assert abs(total - float(z['total_loss'])) / float(z['total_loss']) < ltol
    assert np.allclose(head_loss.numpy(), z['head_losses'], rtol=ltol * 5, atol=1e-5)
    if precision != 'bf16':
        assert np.allclose(head_acc.numpy(), z['head_acc'], atol=1e-6)


def _grads_vs_golden(precision, tol_named, tol_norm):
    from pianobart_amd import ops
    z = np.load(os.path.join(GOLD, 'g4_grads_small.npz'))
    m = _lm(64, 64, 2, 128, 4, 23, precision, dropout=0.0).train().cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(2, 64, seed=9)]
    return z, m, (enc, dec, loss_mask, emask, dmask, target)


@pytest.mark.parametrize('precision,tol_named,tol_norm', [('fp32', 1e-3, 2e-3), ('bf16x3', 1e-3, 2e-3), ('bf16', 1e-1, 4e-2)])       # measured: 4.7e-6 / 2.8e-6, 7.7e-5 / 1.4e-5, 4.8e-2 / 1.3e-2
def test_g4_backward_via_autograd_module_path(precision, tol_named, tol_norm):
    """Drop-in path: PianoBartLM.forward -> list of 8 tensors -> reference-style loss -> .backward()."""
    _need_gpu()
    from oracle import pianobart_oracle as O          # checker only: the loss formula of pretrain.py:112-118,185-189
    z, m, (enc, dec, loss_mask, emask, dmask, target) = _grads_vs_golden(precision, tol_named, tol_norm)
    y = m(enc, dec, emask, dmask)
    total, *_ = O.pretrain_loss(y, target, loss_mask, E2W)
    assert abs(float(total) - float(z['total_loss'])) / float(z['total_loss']) < {'fp32': 1e-4, 'bf16x3': 1e-4, 'bf16': 2e-2}[precision]
    m.zero_grad()
    total.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads.keys()) == list(z['param_names'])        # same set of parameters receives gradient
    worst = 0.0
    for k in z.files:
        if k.startswith('grad__'):
            r = _rel(grads[k[6:]], torch.from_numpy(z[k]))
            worst = max(worst, r)
            assert r < tol_named, (k, r)
    print('g4 %s: worst named gradient %.2e' % (precision, worst))
    norms = np.array([float(grads[k].double().norm()) for k in z['param_names']])
    # k_proj.bias gradients are mathematically zero (softmax shift invariance): absolute floor relative to the typical norm
    print('g4 %s: worst per-parameter norm error %.2e (of norm + 0.1 median)' % (precision, float(np.max(np.abs(norms - z['per_param_grad_norm']) / (z['per_param_grad_norm'] + 1e-1 * np.median(z['per_param_grad_norm']))))))
    bad = np.abs(norms - z['per_param_grad_norm']) > tol_norm * z['per_param_grad_norm'] + tol_norm * 1e-1 * np.median(z['per_param_grad_norm'])
    assert not bad.any(), [(z['param_names'][i], norms[i], z['per_param_grad_norm'][i]) for i in np.nonzero(bad)[0][:5]]


def test_bf16x3_step_is_reproducible_bit_for_bit():
    """The split-bf16 instantiation has no atomics left since its embedding-table gradient runs as Onehot^T dz_hi + Onehot^T dz_lo on the matrix cores (round 6; the exact-f32
    instantiation scatters with f32 atomics, whose order changes from run to run): the same batch twice gives the same loss and the same gradient buffer to the last bit."""
    _need_gpu()
    from pianobart_amd import ops
    z, m, (enc, dec, loss_mask, emask, dmask, target) = _grads_vs_golden('bf16x3', 1e-3, 2e-3)
    eng = m._get_engine()
    eng.bind(enc.device)
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    outs = []
    for _ in range(3):
        eng.zero_accumulated_grads()
        r = eng.loss_and_grads(*args, train=True)
        torch.cuda.synchronize()
        outs.append((r.clone(), torch.cat([g.reshape(-1).clone() for g in eng.grad_views])))
    assert (enc.shape[0] * enc.shape[1]) % 64 == 0                      # the shape takes the one-hot route
    assert torch.isfinite(outs[0][1]).all() and float(outs[0][1].abs().max()) > 0
    for sums, g in outs[1:]:
        assert torch.equal(sums, outs[0][0]) and torch.equal(g, outs[0][1])


def test_g4_fused_step_matches_golden_adamw():
    """Fused engine path: loss_and_grads + clip + HF AdamW vs the golden post-step checksums (fp32)."""
    _need_gpu()
    from pianobart_amd import ops
    z, m, (enc, dec, loss_mask, emask, dmask, target) = _grads_vs_golden('fp32', 0, 0)
    eng = m._get_engine()
    eng.bind(enc.device)
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    eng.loss_and_grads(ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask, train=True)
    named = dict(m.named_parameters())
    gviews = dict(zip([id(p) for p in eng.params], eng.grad_views))
    gn = torch.sqrt(sum((gviews[id(named[k])].double() ** 2).sum() for k in z['param_names'] if not k.endswith('decoder_linear.weight') and not k.endswith('decoder_linear.bias')))
    assert abs(float(gn) - float(z['grad_norm'])) / float(z['grad_norm']) < 1e-3
    eng.optimizer_step(lr=2e-5)
    torch.cuda.synchronize()
    names = list(z['step_param_names'])
    delta = np.array([float((named[k].detach() - before[k]).double().norm()) for k in names])
    psum = np.array([float(named[k].detach().double().sum()) for k in names])
    assert np.allclose(delta, z['adamw_delta_norm'], rtol=2e-3, atol=1e-9)
    assert np.allclose(psum, z['adamw_param_sum'], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('lr', [2e-5, 1e-3])
def test_five_training_steps_follow_the_oracle(lr, precision):
    """pretrain.py:159-196 five times over (fresh batch each step, dropout 0): forward, masked 8-head CE, backward, clip at 3.0, HF AdamW
    with its moments carried from step to step -- the fused HIP loop (pipelined parameter update included) ends on the oracle's
    parameters. lr 1e-3 makes the steps large enough that a wrong bias correction or decay order would show."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    from pianobart_amd import ops
    m = _lm(48, 64, 2, 128, 4, 61, precision, dropout=0.0).train()
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=48, d_model=64, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=128,
                                               decoder_ffn_dim=128, encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0), E2W, W2E)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    eng.pipeline_updates = True
    params = [p for p in o.parameters()]
    live = mo = vo = None
    for step in range(1, 6):
        enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(3, 48, seed=300 + step)
        o.zero_grad()
        tot, *_ = O.pretrain_loss(o(enc, dec, emask, dmask), target, loss_mask, E2W)
        tot.backward()
        if live is None:
            live = [p for p in params if p.grad is not None]
            mo = [torch.zeros_like(p) for p in live]; vo = [torch.zeros_like(p) for p in live]
        grads = [p.grad for p in live]
        O.clip_grad_norm(grads, 3.0)
        with torch.no_grad():
            O.hf_adamw_step([p.data for p in live], grads, mo, vo, step=step, lr=lr)
        g = [t.cuda() for t in (enc, dec, loss_mask, emask, dmask, target)]
        sums = eng.loss_and_grads(ops.ids_to_i16(g[0]), ops.ids_to_i16(g[1]), ops.ids_to_i16(g[5]), g[2].contiguous(), g[3], g[4], train=True)
        eng.optimizer_step(lr=lr)
        s = sums.double().cpu()
        w = torch.tensor(O.loss_weights(E2W), dtype=torch.double)
        assert abs(float(((s[0:8] / s[8:16]) * w).sum() / w.sum()) - float(tot.detach())) / float(tot.detach()) < 2e-4, step
    eng.finish_updates()
    torch.cuda.synchronize()
    po = dict(o.named_parameters())
    worst, worst_k, worst_zero = 0.0, '', 0.0
    for k, p in m.named_parameters():
        if po[k].grad is None:
            continue
        r = _rel(p.detach(), po[k].detach())
        # k_proj.bias: its gradient is mathematically zero (softmax shift invariance), what arrives is rounding noise, and AdamW divides by sqrt(v) + 1e-6:
        # noise of 1e-9 (exact f32) moves the bias by 1e-3 lr per step, noise of 1e-7 .. 1e-6 (split bf16) by a good part of lr -- in a direction the oracle's own
        # noise does not share. Reported, bounded by the 5 lr such a walk can cover, and kept out of the bound on the parameters that have a gradient.
        if k.endswith('k_proj.bias'):
            worst_zero = max(worst_zero, float((p.detach().cpu().double() - po[k].detach().double()).abs().max()))
            continue
        if r > worst:
            worst, worst_k = r, k
    print('five steps %s lr %g: worst parameter %.2e (%s); k_proj.bias walk %.2e (5 lr = %.0e)' % (precision, lr, worst, worst_k, worst_zero, 5 * lr))
    assert worst_zero <= 5 * lr * 1.01
    # AdamW's update lr g / (|g| + 1e-6) turns an ABSOLUTE gradient error d into lr d / 1e-6 on the elements whose gradient is far below its eps: the exact-f32
    # instantiation's 1e-9 moves such an element by 1e-3 lr per step, the split-bf16 one's 1e-7 by a tenth of lr (measured worst tensor, relative to its largest
    # element: 9e-5 at lr 2e-5, 1.8e-2 at lr 1e-3, both in the 2048 -> d merge Linear) -- while every step's loss above agrees to 2e-4 and the 200-step loss curve
    # to 1e-3 over 150 steps (profiles/r06_loss_overlay_bf16x3.txt): those elements are the directions the loss does not feel.
    bound = (2e-5 if lr < 1e-4 else 2e-3) if precision == 'fp32' else (3e-4 if lr < 1e-4 else 5e-2)
    assert worst < bound, (worst, worst_k)


def test_dropout_train_step_is_consistent():
    """Dropout active (p=0.1): forward mask == backward mask. Check by finite differences on one bias."""
    _need_gpu()
    from pianobart_amd import ops
    m = _lm(64, 64, 2, 128, 4, 23, 'fp32', dropout=0.1).train().cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(2, 64, seed=9)]
    eng = m._get_engine()
    eng.bind(enc.device)
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)

    def loss_at(seed_state):
        eng._seed = seed_state
        s = eng.loss_and_grads(*args, train=True).cpu().double()
        return float(((s[0:8] / s[8:16]) * w).sum() / w.sum())

    base = loss_at(77)
    p = m.pianobart.bart.decoder.layers[0].fc1.bias
    g = eng.grad_views[[id(q) for q in eng.params].index(id(p))].clone()
    idx = int(g.abs().argmax())
    eps = 1e-2
    with torch.no_grad():
        p[idx] += eps
    up = loss_at(77)
    with torch.no_grad():
        p[idx] -= 2 * eps
    dn = loss_at(77)
    fd = (up - dn) / (2 * eps)
    assert abs(fd - float(g[idx])) < 0.05 * abs(float(g[idx])) + 1e-5, (fd, float(g[idx]))
    assert abs(loss_at(78) - base) > 1e-6          # a different seed gives a different mask


def test_g8_generate_trace():
    _need_gpu()
    z = np.load(os.path.join(GOLD, 'g8_generate.npz'))
    m = _lm(24, 64, 2, 128, 4, 31, 'fp32').eval()
    assert sd_checksum(m.state_dict()) == _sha(z)
    m = m.cuda()
    enc = torch.from_numpy(z['enc']).long().cuda(); emask = torch.from_numpy(z['emask']).cuda()
    np.random.seed(2023)
    out = m(enc, None, emask, None, generate=True, device_num=0)
    assert out.shape == (1, 24, 8) and out.dtype == torch.int64
    assert np.array_equal(out.cpu().numpy(), z['tokens'])


@pytest.mark.parametrize('seed', list(range(8)))
def test_generate_against_the_oracle_on_random_prompts(seed):
    """model.py:28-107 beyond the one golden trace: random weights, random prompts of every length (one row ... the full window), the
    global np.random stream seeded alike on both sides -- the HIP decode (K/V caches, one token per step) emits the oracle's tokens
    (encoder + decoder re-run per position, as the reference does), early stop and PAD fill included, and leaves the RNG where the
    oracle leaves it."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    S = 20
    m = _lm(S, 64, 1 + seed % 2, 128, 2 + 2 * (seed % 2), 50 + seed, 'fp32').eval()
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=S, d_model=64, encoder_layers=1 + seed % 2, decoder_layers=1 + seed % 2,
                                               encoder_ffn_dim=128, decoder_ffn_dim=128, encoder_attention_heads=2 + 2 * (seed % 2),
                                               decoder_attention_heads=2 + 2 * (seed % 2)), E2W, W2E)).eval()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    L = [1, 2, 5, 9, 13, 17, 19, 20][seed]
    enc = synth_octuple_batch(1, S, seed=200 + seed, min_len=L)[5]
    pad = torch.from_numpy(m.pianobart.pad_word_np)
    enc[0, L:] = pad                                                       # a prompt of exactly L rows
    emask = (enc[:, :, 0] != int(m.pianobart.bar_pad_word)).float()
    with torch.no_grad():
        np.random.seed(777 + seed)
        want = o(enc, None, emask, None, generate=True, device_num=-1)
        st_o = np.random.get_state()[1].copy()
        np.random.seed(777 + seed)
        got = m(enc.cuda(), None, emask.cuda(), None, generate=True, device_num=0)
        st_m = np.random.get_state()[1].copy()
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want.numpy())
    assert np.array_equal(st_o, st_m)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_g10_cfg2_shape_spot_check(precision):
    """cfg-2 model shape (12L/768/ffn3072/12 heads, S=1024, B=1) against vectors captured from the reference: both parity-grade
    instantiations -- exact f32 and the split-bf16 GEMMs (round 6) -- under the north-star bound (logits 1e-3, argmax identical outside near-ties)."""
    _need_gpu()
    z = np.load(os.path.join(GOLD, 'g10_cfg2_spot.npz'))
    m = _lm(1024, 768, 12, 3072, 12, 41, precision).eval()
    assert sd_checksum(m.state_dict()) == _sha(z)
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(1, 1024, seed=19)]
    with torch.no_grad():
        y = torch.cat(m(enc, dec, emask, dmask), dim=-1)[0].cpu()
    rows = z['rows']
    rel = float((y[rows] - torch.from_numpy(z['logit_rows'])).abs().max() / float(z['logit_absmax']))
    print('cfg2 logits rel (%s) = %.3e' % (precision, rel))
    assert rel < 1e-3
    offs = np.cumsum([0, 262, 134, 135, 262, 134, 38, 260, 55])
    arg = torch.stack([y[:, offs[i]:offs[i + 1]].argmax(-1) for i in range(8)], dim=-1).numpy()
    clear = z['top2_gap'] > 1e-4 * float(z['logit_absmax'])
    assert np.array_equal(arg[clear], z['argmax'].astype(np.int64)[clear]) and clear.mean() > 0.99


def test_cpu_tensors_fail_loudly():
    from pianobart_amd._lib import PBError
    m = _lm(24, 64, 1, 128, 4, 1, 'fp32').eval()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(1, 24, seed=1)
    with pytest.raises(PBError):
        m(enc, dec, emask, dmask)


@pytest.mark.parametrize('precision,d,heads', [('fp32', 128, 4), ('bf16', 128, 4), ('fp32', 192, 2), ('bf16', 192, 2), ('bf16', 256, 4)])
def test_generate_kv_cache_equals_full_rerun(precision, d, heads):
    """KV-cached decode (encoder once, cross K/V once, one token per step) == re-running the decoder over all positions
    (head_dim 32, 96 and 64 through the native single-query attention kernel)."""
    _need_gpu()
    m = _lm(48, d, 2, 256, heads, 77, precision).eval()
    with torch.no_grad():                                   # make special tokens unsamplable: the decode runs all 48 steps
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            m.mask_lm.proj[i].bias[p0:] = -30.0
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(1, 48, seed=4, min_len=40)]
    eng = m._get_engine()
    np.random.seed(5)
    a = eng.generate(enc, emask, m.sample_row, use_cache=True)
    np.random.seed(5)
    b = eng.generate(enc, emask, m.sample_row, use_cache=False)
    np.random.seed(5)
    c = eng._generate_pyloop(enc, emask, m.sample_row)
    assert torch.equal(b, c)                                  # python-sequenced cache == full re-run (same kernels, bitwise)
    if precision == 'fp32':
        assert torch.equal(a, b)                              # native GEMV path: same tokens
    # teacher-forced: feed the SAME token sequence to the native step and to the python-sequenced cache and compare the
    # logits row of every step (sampled traces of the bf16 paths may legitimately part ways after one near-tie sample)
    forced = b[0].cpu()
    def recorder(store):
        def fn(row):
            store.append(row.clone())
            return forced[len(store) - 1].clone()
        return fn
    la, lc = [], []
    eng.generate(enc, emask, recorder(la), use_cache=True)
    eng._generate_pyloop(enc, emask, recorder(lc))
    assert len(la) == len(lc) == 48
    tol = 1e-4 if precision == 'fp32' else 3e-2               # bf16: GEMV (f32 accumulate over K in lane order) vs MFMA tiles
    for i, (x, y) in enumerate(zip(la, lc)):
        keep = y > -20                                         # the -30 biased special tokens carry no information
        assert _rel(x[keep], y[keep]) < tol, (i, _rel(x[keep], y[keep]))
    assert int((a[0, :, 0] != 256).sum()) == 48         # every position was generated


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-4), ('bf16', 8e-2)])
def test_head_dim_96(precision, tol):
    """Reference CLI default heads=8 gives head_dim 96 at d=768: fused attention with a half-filled second [64][64] image (bf16);
    GEMM + masked softmax in the exact-f32 instantiation."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    m = _lm(40, 192, 1, 256, 2, 5, precision, dropout=0.0).train()
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=40, d_model=192, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=256,
                                               decoder_ffn_dim=256, encoder_attention_heads=2, decoder_attention_heads=2, dropout=0.0), E2W, W2E)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(2, 40, seed=6)
    total_o, *_ = O.pretrain_loss(o(enc, dec, emask, dmask), target, loss_mask, E2W)
    total_o.backward()
    y = m(enc.cuda(), dec.cuda(), emask.cuda(), dmask.cuda())
    total, *_ = O.pretrain_loss(y, target.cuda(), loss_mask.cuda(), E2W)
    total.backward()
    assert abs(float(total) - float(total_o)) / float(total_o) < tol
    go = dict(o.named_parameters()); gm = dict(m.named_parameters())
    for k in ('pianobart.bart.encoder.layers.0.self_attn.q_proj.weight', 'pianobart.bart.decoder.layers.0.encoder_attn.v_proj.weight', 'pianobart.word_emb.3.lut.weight'):
        assert _rel(gm[k].grad, go[k].grad) < (1e-3 if precision != 'bf16' else 0.2), k


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16x3', 1e-4), ('bf16', 6e-2)])
def test_ragged_length_and_fully_padded_sample(precision, tol):
    """S = 200 (not a multiple of the 64/128-row attention tiles), head_dim 64 (flash64 path in bf16), one sample that is
    PAD from the first row on (every encoder key masked -> zero-row rule), B = 3: logits, loss and grad norm vs the oracle."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    kw = dict(max_position_embeddings=200, d_model=256, encoder_layers=1, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
              encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0)
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    m = PianoBartLM(PianoBart(BartConfig(**kw), E2W, W2E, precision=precision)).train()
    randomize_params(m, 21)
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**kw), E2W, W2E)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(3, 200, seed=14)
    pad = torch.tensor([256, 128, 129, 256, 128, 32, 254, 49])
    enc[2] = pad; target[2] = pad; dec[2, 1:] = pad                      # sample 2: nothing but PAD (decoder keeps its SOS row)
    emask = (enc[:, :, 0] != 256).float(); dmask = (dec[:, :, 0] != 256).float()
    assert float(emask[2].sum()) == 0
    yo = o(enc, dec, emask, dmask)
    total_o, *_ = O.pretrain_loss(yo, target, loss_mask, E2W)
    total_o.backward()
    y = m(enc.cuda(), dec.cuda(), emask.cuda(), dmask.cuda())
    assert _rel(torch.cat(y, -1), torch.cat(yo, -1).detach()) < tol
    assert not torch.isnan(torch.cat(y, -1)).any()
    total, *_ = O.pretrain_loss(y, target.cuda(), loss_mask.cuda(), E2W)
    total.backward()
    assert abs(float(total) - float(total_o)) / float(total_o) < tol
    gn = lambda mod: float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in mod.parameters() if p.grad is not None)))
    assert abs(gn(m) - gn(o)) / gn(o) < (1e-3 if precision != 'bf16' else 5e-2)


def test_tiny_batch_one_short_sequence():
    _need_gpu()
    from oracle import pianobart_oracle as O
    m = _lm(8, 64, 1, 64, 2, 9, 'fp32').eval()
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=8, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=64,
                                               decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2), E2W, W2E)).eval()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(1, 8, seed=1, min_len=5)
    with torch.no_grad():
        assert _rel(torch.cat(m(enc.cuda(), dec.cuda(), emask.cuda(), dmask.cuda()), -1), torch.cat(o(enc, dec, emask, dmask), -1)) < 1e-4


def test_id_outside_its_table_raises_index_error_like_nn_embedding():
    """PianoBart.py:15-16: nn.Embedding raises IndexError on an id >= its table size (and the reference's CPU path on a negative one).
    The HIP route checks on the device and raises at the forward's own synchronisation point; valid batches are untouched."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    m = _lm(32, 64, 1, 64, 2, 3, 'fp32').eval().cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(2, 32, seed=2)]
    with torch.no_grad():
        m(enc, dec, emask, dmask)                                        # clean batch: no error
        for col, bad in ((1, 134), (5, 38), (0, -1), (3, 70000)):
            e = enc.clone(); e[1, 7, col] = bad
            with pytest.raises(IndexError):
                m(e, dec, emask, dmask)
            d = dec.clone(); d[0, 3, col] = bad
            with pytest.raises(IndexError):
                m(enc, d, emask, dmask)
        m(enc, dec, emask, dmask)                                        # the mark does not stick
        o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=32, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=64,
                                                   decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2), E2W, W2E)).eval()
        e = enc.cpu().clone(); e[1, 7, 1] = 134
        with pytest.raises(IndexError):
            o(e, dec.cpu(), emask.cpu(), dmask.cpu())                    # the checker agrees on the error type


def test_missing_masks_and_other_mask_dtypes():
    """PianoBart.forward's masks default to None (= everything visible, modeling_bart.py) and callers hand over float, bool or integer
    0 / 1 masks: same logits as the oracle for every spelling."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    m = _lm(48, 64, 2, 128, 4, 13, 'fp32').eval()
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=48, d_model=64, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=128,
                                               decoder_ffn_dim=128, encoder_attention_heads=4, decoder_attention_heads=4), E2W, W2E)).eval()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(3, 48, seed=4)
    with torch.no_grad():
        ref_none = torch.cat(o(enc, dec, None, None), -1)
        assert _rel(torch.cat(m(enc.cuda(), dec.cuda()), -1), ref_none) < 1e-4
        assert _rel(torch.cat(m(enc.cuda(), dec.cuda(), None, dmask.cuda()), -1), torch.cat(o(enc, dec, None, dmask), -1)) < 1e-4
        ref = torch.cat(o(enc, dec, emask, dmask), -1)
        for cast in (lambda t: t.bool(), lambda t: t.long(), lambda t: t.to(torch.int32), lambda t: t.double(), lambda t: t.half()):
            assert _rel(torch.cat(m(enc.cuda(), dec.cuda(), cast(emask).cuda(), cast(dmask).cuda()), -1), ref) < 1e-4
        # int32 / int16 ids instead of int64 (the loader's shards)
        assert _rel(torch.cat(m(enc.int().cuda(), dec.short().cuda(), emask.cuda(), dmask.cuda()), -1), ref) < 1e-4


@pytest.mark.parametrize('S,d,Le,Ld,fe,fd,h,B', [
    (8, 64, 1, 1, 64, 64, 2, 1),            # head_dim 32, one short window
    (24, 96, 1, 2, 160, 96, 1, 3),          # one head of 96; decoder deeper than the encoder; ffn not a multiple of 64
    (40, 128, 2, 1, 200, 328, 4, 2),        # ffn sizes that are multiples of 8 only
    (100, 192, 1, 1, 384, 384, 2, 2),       # head_dim 96
    (130, 256, 1, 1, 512, 256, 2, 2),       # head_dim 128, S not a multiple of 64
    (57, 320, 1, 1, 640, 640, 5, 3),        # d = 320, 5 heads
    (33, 64, 1, 1, 96, 96, 4, 2),           # head_dim 16
    (77, 48, 1, 1, 72, 72, 2, 2),           # head_dim 24, d = 48
])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
def test_shape_sweep_forward_loss_and_gradients_against_the_oracle(S, d, Le, Ld, fe, fd, h, B, precision):
    """Shapes off the beaten path (odd sequence lengths, head_dim 16 ... 128, ffn sizes that are not tile multiples, unequal encoder /
    decoder depth and width): the exact-f32 instantiation agrees with the oracle on logits, loss and every parameter gradient (the bf16
    one within its looser bound: logits 5e-2, loss 2e-3, gradient norm 2e-2; the split-bf16 one within 2e-4 / 1e-5 / 1e-4 and every gradient 2e-3) -- or the library says loudly that it does not cover the
    shape; never a silent difference."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from pianobart_amd._lib import PBError
    kw = dict(max_position_embeddings=S, d_model=d, encoder_layers=Le, decoder_layers=Ld, encoder_ffn_dim=fe, decoder_ffn_dim=fd,
              encoder_attention_heads=h, decoder_attention_heads=h, dropout=0.0)
    m = PianoBartLM(PianoBart(BartConfig(**kw), E2W, W2E, precision=precision))
    randomize_params(m, 5)
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**kw), E2W, W2E)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.train().cuda()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(B, S, seed=S + d, min_len=max(2, S // 3))
    try:
        y = m(enc.cuda(), dec.cuda(), emask.cuda(), dmask.cuda())
        tot, *_ = O.pretrain_loss(y, target.cuda(), loss_mask.cuda(), E2W)
        tot.backward()
    except (PBError, RuntimeError) as e:
        assert 'pb_' in str(e) or 'pianobart' in str(e).lower(), e         # a refusal must name its origin
        pytest.skip('shape refused loudly: %s' % str(e)[:120])
    yo = o(enc, dec, emask, dmask)
    tot_o, *_ = O.pretrain_loss(yo, target, loss_mask, E2W)
    tot_o.backward()
    f32 = precision == 'fp32'
    tl, tloss, tnorm, tgrad = {'fp32': (1e-4, 1e-4, 1e-3, 2e-3), 'bf16x3': (2e-4, 1e-5, 1e-4, 2e-3), 'bf16': (BF16_LOGITS, BF16_LOSS, BF16_NORM, None)}[precision]
    e_l = _rel(torch.cat(y, -1).detach(), torch.cat(yo, -1).detach())
    e_loss = abs(float(tot.detach()) - float(tot_o.detach())) / float(tot_o.detach())
    go = {k: p.grad for k, p in o.named_parameters() if p.grad is not None}
    gm = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(go) == sorted(gm)
    n_o = float(torch.sqrt(sum((g.double() ** 2).sum() for g in go.values())))
    n_m = float(torch.sqrt(sum((g.double().cpu() ** 2).sum() for g in gm.values())))
    e_n = abs(n_m - n_o) / n_o
    print('sweep %s S=%d d=%d h=%d: logits %.2e loss %.2e grad-norm %.2e' % (precision, S, d, h, e_l, e_loss, e_n))
    assert e_l < tl and e_loss < tloss and e_n < tnorm, (e_l, e_loss, e_n)
    if tgrad is None:
        return
    scale = max(float(g.abs().max()) for g in go.values())
    for k, g in go.items():
        err = float((gm[k].cpu().double() - g.double()).abs().max())
        assert err < tgrad * max(float(g.abs().max()), 1e-3 * scale), (k, err, float(g.abs().max()))


def test_head_without_a_loss_position_is_nan_like_the_reference():
    """pretrain.py:117 divides a head's masked loss sum by its mask count: a head with no loss position in the whole batch is 0 / 0,
    the total loss and every gradient that hangs on that head's logits become NaN. Same on the HIP path, fused and
    module route; the other heads' numbers are untouched."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    from pianobart_amd import ops
    m = _lm(64, 64, 1, 128, 2, 17, 'fp32', dropout=0.0).train()
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(max_position_embeddings=64, d_model=64, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=128,
                                               decoder_ffn_dim=128, encoder_attention_heads=2, decoder_attention_heads=2, dropout=0.0), E2W, W2E)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(2, 64, seed=21)
    loss_mask = loss_mask.clone()
    loss_mask[:, :, 3] = 0                                               # the Pitch head never carries a loss term
    tot_o, head_o, *_ = O.pretrain_loss(o(enc, dec, emask, dmask), target, loss_mask, E2W)
    tot_o.backward()
    assert torch.isnan(tot_o) and torch.isnan(o.mask_lm.proj[3].weight.grad).all()
    g = [t.cuda() for t in (enc, dec, loss_mask, emask, dmask, target)]
    eng = m._get_engine()
    eng.bind(g[0].device)
    sums = eng.loss_and_grads(ops.ids_to_i16(g[0]), ops.ids_to_i16(g[1]), ops.ids_to_i16(g[5]), g[2].contiguous(), g[3], g[4], train=True).cpu().double()
    head = sums[0:8] / sums[8:16]
    assert float(sums[8 + 3]) == 0.0 and torch.isnan(head[3])
    keep = [i for i in range(8) if i != 3]
    ho = torch.stack([h.detach().double() for h in head_o]) if isinstance(head_o, (list, tuple)) else head_o.detach().double()
    assert torch.allclose(head[keep], ho[keep], rtol=1e-4, atol=1e-6)
    torch.cuda.synchronize()
    assert torch.isnan(eng.g['head.w']).any()                            # the NaN reaches the head weights (and everything below them)
    y = m(g[0], g[1], g[3], g[4])
    tot_m, *_ = O.pretrain_loss(y, g[5], g[2], E2W)
    assert torch.isnan(tot_m)


@pytest.mark.gpu
def test_batch_without_any_loss_position_is_nan_like_the_reference_also_under_row_packing():
    """No loss position in ANY head (pretrain.py:116-117: 0 / 0 eight times): the reference's loss and every parameter gradient are NaN.
    The packed step would run its last layer on zero rows and leave zero gradients; rowpack.pack_batch sends such a batch down the
    padded step, so the bf16 engine at a packable shape (head_dim 64, rows to drop) poisons the step exactly like the reference."""
    _need_gpu()
    from oracle import pianobart_oracle as O
    from pianobart_amd import ops
    kw = dict(max_position_embeddings=256, d_model=128, encoder_layers=1, decoder_layers=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
              encoder_attention_heads=2, decoder_attention_heads=2, dropout=0.0)
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**kw), E2W, W2E)).train()
    randomize_params(o, 5)
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(4, 256, seed=33)
    loss_mask = torch.zeros_like(loss_mask)
    tot_o, *_ = O.pretrain_loss(o(enc, dec, emask, dmask), target, loss_mask, E2W)
    tot_o.backward()
    assert torch.isnan(tot_o) and all(torch.isnan(p.grad).all() for p in o.mask_lm.parameters())
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    m = PianoBartLM(PianoBart(BartConfig(**kw), E2W, W2E, precision='bf16')).train()
    m.load_state_dict(o.state_dict(), strict=True)
    m = m.cuda()
    eng = m._get_engine()
    g = [t.cuda() for t in (enc, dec, loss_mask, emask, dmask, target)]
    eng.bind(g[0].device)
    assert float(emask.mean()) < 0.95                                    # rows to drop: with a loss position somewhere this batch WOULD be packed
    sums = eng.loss_and_grads(ops.ids_to_i16(g[0]), ops.ids_to_i16(g[1]), ops.ids_to_i16(g[5]), g[2].contiguous(), g[3], g[4], train=True).cpu()
    assert eng.last_rows[0] == 4 * 256                                   # the padded step ran
    assert float(sums[8:16].sum()) == 0.0
    torch.cuda.synchronize()
    for name in ('head.w', 'dec.0.wqkv', 'enc.0.w1', 'dec.wkv_all', 'emb'):
        assert torch.isnan(eng.g[name]).all(), name


@pytest.mark.gpu
def test_fused_step_checks_caller_supplied_ids():
    """Engine.loss_and_grads on ids it did not generate (PianoBart.py:15-16: nn.Embedding raises IndexError on an id outside its table):
    the ids are range-checked on the device, an offending id never reaches a gather, and IndexError is raised where the host learns the
    verdict -- inside the call when the packing plan waits for its row counts anyway. ids_checked=True (the bench's generated ids, the
    Pretrainer's validated host batch) skips the check."""
    _need_gpu()
    from pianobart_amd import ops
    m = _lm(128, 128, 1, 128, 2, 9, 'bf16', dropout=0.0).train().cuda()
    eng = m._get_engine()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(4, 128, seed=12)]
    eng.bind(enc.device)
    args = lambda e, d: (ops.ids_to_i16(e), ops.ids_to_i16(d), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    eng.loss_and_grads(*args(enc, dec), train=True)                       # clean batch
    torch.cuda.synchronize()
    eng._raise_if_bad_ids()
    for col, bad in ((1, 134), (3, -5)):
        e = enc.clone(); e[1, 7, col] = bad
        with pytest.raises(IndexError):
            eng.loss_and_grads(*args(e, dec), train=True)
            torch.cuda.synchronize()
            eng._raise_if_bad_ids()                                      # (a step that did not wait for anything learns it here at the latest)
        if bad > 0:                                                       # the caller's own int16 tensor is checked on a copy: never rewritten
            mine = ops.ids_to_i16(e); keep = mine.clone()
            with pytest.raises(IndexError):
                eng.loss_and_grads(mine, *args(e, dec)[1:], train=True)
                torch.cuda.synchronize()
                eng._raise_if_bad_ids()
            assert torch.equal(mine, keep) and int(mine[1, 7, col]) == bad
        d = dec.clone(); d[2, 5, col] = bad
        with pytest.raises(IndexError):
            eng.loss_and_grads(*args(enc, d), train=True)
            torch.cuda.synchronize()
            eng._raise_if_bad_ids()
    sums = eng.loss_and_grads(*args(enc, dec), train=True)                # the mark does not stick
    torch.cuda.synchronize()
    eng._raise_if_bad_ids()
    assert torch.isfinite(sums).all()


@pytest.mark.parametrize('precision,train', [('bf16', True), ('bf16', False), ('fp32', True)])
def test_second_stream_gives_identical_gradients(precision, train):
    """Weight-gradient GEMMs and the cross-attention K/V projections run on a second HIP stream (engine._WGRAD_STREAM): every
    gradient must be bit-identical to the one-stream schedule, over repeated steps (a missed wait shows up as a stale operand).
    train=False is the p=0 schedule, where the LayerNorm backward's output is itself the weight-gradient operand."""
    _need_gpu()
    from pianobart_amd import engine as E, ops
    m = _lm(256, 256, 2, 512, 4, 31, precision, dropout=0.1).train().cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(4, 256, seed=3)]
    eng = m._get_engine()
    eng.bind(enc.device)
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    saved = E._WGRAD_STREAM, E._WG_TARGET

    def grads(mode):
        E._WGRAD_STREAM = mode
        out = []
        for it in range(3):
            eng._seed = 1234 + it
            s = eng.loss_and_grads(*args, train=train)
            torch.cuda.synchronize()
            out.append((eng.G32.clone(), s.clone()))
        return out

    try:
        E._WG_TARGET = 256
        one = grads(0)
        two = grads(7)
    finally:
        E._WGRAD_STREAM, E._WG_TARGET = saved
    if eng._side_stream() is None:
        pytest.skip('no second stream that runs concurrently on this box')
    # the exact-f32 route scatter-adds the embedding-table / position gradients with f32 atomics (order varies run to run)
    atomic = ('emb', 'lin.w', 'enc.pos', 'dec.pos') if precision == 'fp32' else ()
    for (g0, s0), (g1, s1) in zip(one, two):
        assert torch.equal(s0, s1)
        for name, sl in eng.slots.items():
            a, b = g0[sl.off:sl.off + sl.numel], g1[sl.off:sl.off + sl.numel]
            if name in atomic:
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), name
            else:
                assert torch.equal(a, b), (name, float((a - b).abs().max()))


def test_backward_after_an_intervening_forward_fails_loudly():
    """The engine keeps the activations of the most recent forward only: a backward through an older graph (here: after a generate
    call reused the workspace) must raise, not return gradients computed from overwritten activations."""
    _need_gpu()
    from pianobart_amd._lib import PBError
    m = _lm(64, 64, 2, 128, 4, 5, 'fp32', dropout=0.0).train().cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(1, 64, seed=2)]
    out = m(enc, dec, emask, dmask)
    loss = sum(o.float().sum() for o in out)
    with torch.no_grad():
        m.eval()
        np.random.seed(0)
        m(enc, encoder_attention_mask=emask, generate=True)
        m.train()
    with pytest.raises(PBError):
        loss.backward()


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16', 4e-2)])
def test_decode_at_cfg2_size_matches_teacher_forced_full_pass(precision, tol):
    """BASELINE configs[3] shape: 12L / 768 / 12 heads, S = 1024, B = 1. 96 decode steps of the KV-cached native path (self-attention
    keys split over workgroups from 65 keys on, cross-attention over ~700 encoder keys in 11 splits, merged in the out-projection
    GEMV) against ONE teacher-forced full decoder pass over the same tokens (the training kernels): the logits row of every step,
    and -- exact-f32 instantiation -- its argmax ids."""
    _need_gpu()
    m = _lm(1024, 768, 12, 3072, 12, 41, precision).eval()
    with torch.no_grad():
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            m.mask_lm.proj[i].bias[p0:] = -30.0                   # specials unsamplable: the loop runs as long as we feed it
    m = m.cuda()
    enc = synth_octuple_batch(1, 1024, seed=19, min_len=700)[5].cuda()
    emask = (enc[:, :, 0] != 256).float()
    assert 600 < int(emask.sum()) <= 1024
    N = 96
    forced = synth_octuple_batch(1, N, seed=23, min_len=N)[5][0]
    forced[-1] = forced[-2]                                       # no EOS row inside the forced prefix
    rows = []

    def feed(row):
        rows.append(row.clone())
        return forced[len(rows) - 1].clone() if len(rows) <= N else torch.tensor([256, 128, 129, 256, 128, 32, 254, 49])

    eng = m._get_engine()
    out = eng.generate(enc, emask, feed)
    assert len(rows) == N + 1 and torch.equal(out[0, :N].cpu(), forced)
    # teacher-forced: decoder input = SOS + forced[:-1]... one full pass gives the logits of every position at once
    dec = torch.tensor([256, 128, 129, 256, 128, 32, 254, 49]).repeat(1, 1024, 1)
    dec[0, 0] = torch.tensor([258, 130, 131, 258, 130, 34, 256, 51])
    dec[0, 1:N + 1] = forced
    dmask = torch.zeros(1, 1024); dmask[0, :N + 1] = 1
    with torch.no_grad():
        full = torch.cat(m(enc, dec.cuda(), emask, dmask.cuda()), dim=-1)[0].float().cpu()
    worst = 0.0
    for i in range(N + 1):
        keep = full[i] > -20
        worst = max(worst, float((rows[i][keep] - full[i][keep]).abs().max() / full[i][keep].abs().max()))
    print('decode 12L/768 S=1024 (%s): worst logits rel over %d steps = %.2e' % (precision, N + 1, worst))
    assert worst < tol
    if precision == 'fp32':
        offs = np.cumsum([0, 262, 134, 135, 262, 134, 38, 260, 55])
        a = torch.stack([torch.stack([r[offs[k]:offs[k + 1]].argmax() for k in range(8)]) for r in rows])
        b = torch.stack([torch.stack([full[i][offs[k]:offs[k + 1]].argmax() for k in range(8)]) for i in range(N + 1)])
        assert float((a == b).float().mean()) > 0.995


@pytest.mark.parametrize('d,heads,S', [(256, 4, 200), (256, 2, 96), (512, 8, 72), (1024, 8, 40), (768, 12, 130)])
def test_graph_decode_equals_the_per_launch_decode(monkeypatch, d, heads, S):
    """Round 3: one hipGraph replay per token (position in device memory, q / k / v projections fused into the split single-query
    attention, 6 launches per layer) against (a) the same launches issued directly: bit-identical logits rows, and (b) the round-2
    per-launch loop (pb_decode_step): the same tokens fed, logits equal to bf16 rounding. head_dim 64 and 128; S = 200 takes the
    self-attention through several key splits (65+ keys) and the cross-attention through 4."""
    _need_gpu()
    from pianobart_amd import engine as E
    m = _lm(S, d, 2, 512, heads, 31, 'bf16').eval()
    with torch.no_grad():
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            m.mask_lm.proj[i].bias[p0:] = -30.0
    m = m.cuda()
    enc = synth_octuple_batch(1, S, seed=8, min_len=S - 9)[5].cuda()
    emask = (enc[:, :, 0] != 256).float()
    forced = synth_octuple_batch(1, S, seed=23, min_len=S)[5][0]
    forced[-1] = forced[-2]
    eng = m._get_engine()

    def run(mode):
        monkeypatch.setattr(E, '_DECODE_GRAPH', mode)
        rows = []

        def feed(row):
            rows.append(row.clone())
            return forced[len(rows) - 1].clone()
        out = eng.generate(enc, emask, feed)
        assert torch.equal(out[0].cpu(), forced)
        return rows, eng.last_decode

    g, info_g = run(1)
    dr, info_d = run(0)
    old, info_o = run(-1)
    assert info_g is not None and info_g['launches_per_token'] == 6 * 2 + 2 and info_g['tokens'] == S, info_g
    assert info_g['graph'] and not info_d['graph'] and info_o is None
    assert len(g) == len(dr) == len(old) == S
    worst = 0.0
    for i in range(S):
        assert torch.equal(g[i], dr[i]), i                         # a replay runs exactly the launches of the direct form
        keep = old[i] > -20
        worst = max(worst, _rel(g[i][keep], old[i][keep]))
    print('graph decode vs per-launch decode d=%d hd=%d S=%d: worst logits rel %.2e' % (d, d // heads, S, worst))
    assert worst < 2e-2
    # a second prompt through a fresh decoder of the same engine: nothing is left over from the first
    g2, _ = run(1)
    assert all(torch.equal(a, b) for a, b in zip(g, g2))


@pytest.mark.parametrize('d,heads,S,sharp', [(256, 4, 200, 1.0), (768, 12, 130, 1.0), (512, 8, 72, 40.0), (1024, 8, 40, 8.0)])
def test_device_sampled_decode_emits_the_host_loops_tokens(monkeypatch, d, heads, S, sharp):
    """Round 6: model.py:68-107 sampled on the device ahead of the host (8 tokens per graph replay, uniform draws of the whole prompt uploaded
    once), the host following behind through the logged logits rows with the reference code path. Against the per-token host loop
    (PB_DECODE_SPEC=0) on the same prompt and RNG seed: the same tokens, the same early stop, the same np.random state afterwards -- with flat
    distributions (random weights: ~230 candidates under the p = 0.9 heads) and peaked ones (LM head scaled up), with `max_new`, and with
    the device's choice corrupted at every 7th position (decode_fault_period: the rewind path must restore the host's sequence)."""
    _need_gpu()
    from pianobart_amd import engine as E
    m = _lm(S, d, 2, 512, heads, 31, 'bf16').eval()
    with torch.no_grad():
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            m.mask_lm.proj[i].weight.mul_(sharp)
            m.mask_lm.proj[i].bias[p0 + 3:] = -30.0                # EOS stays reachable: some runs stop early by themselves
            m.mask_lm.proj[i].bias[p0:p0 + 3] = -30.0
    m = m.cuda()
    enc = synth_octuple_batch(1, S, seed=8, min_len=S - 9)[5].cuda()
    emask = (enc[:, :, 0] != 256).float()
    eng = m._get_engine()
    sampler = dict(T=m.SAMPLE_T, P=m.SAMPLE_P)

    def run(spec, fault=0, max_new=None, seed=5):
        monkeypatch.setattr(E, '_DECODE_SPEC', spec)
        eng.decode_fault_period = fault
        np.random.seed(seed)
        out = eng.generate(enc, emask, m.sample_row, max_new=max_new, sampler=sampler)
        return out.cpu(), np.random.get_state()[1].copy(), dict(eng.last_decode)

    want, st_w, info_w = run(0)
    got, st_g, info_g = run(1)
    assert not info_w.get('device_sampler') and info_g['device_sampler'] and info_g['graph']
    assert info_g['launches_per_token'] == 6 * 2 + 3 and info_g['tokens'] == info_w['tokens']
    assert torch.equal(got, want) and np.array_equal(st_g, st_w)
    assert info_g['rewinds'] <= 2, info_g                              # the device's own choice is (almost) always the host's
    got_f, st_f, info_f = run(1, fault=7)
    assert torch.equal(got_f, want) and np.array_equal(st_f, st_w)
    assert info_f['rewinds'] >= min(info_w['tokens'], S) // 7 - 1 and info_f['rewinds'] > 0, info_f
    for cut in (1, 8, 13):
        a, sa, ia = run(0, max_new=cut, seed=9)
        b, sb, ib = run(1, max_new=cut, seed=9)
        assert torch.equal(a, b) and np.array_equal(sa, sb) and ia['tokens'] == ib['tokens'] <= cut
    eng.decode_fault_period = 0
    # through the module surface (PianoBartLM.forward(generate=True)): the sampler is named there, so this is the default path
    monkeypatch.setattr(E, '_DECODE_SPEC', 1)
    np.random.seed(5)
    y = m(enc, None, emask, None, generate=True, device_num=-1)
    assert torch.equal(y, want) and eng.last_decode['device_sampler']
