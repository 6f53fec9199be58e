"""GPU: the bf16 throughput kernels AT THE BENCHMARKED SHAPE (BASELINE configs[1]: 12L / 768 / ffn 3072 / 12 heads, S = 1024,
B = 32 -> T = 32768 rows) and at configs[4]'s shape (24L / 1024 / ffn 4096 / 16 heads, S = 2048).

The small-shape tests (test_kernels_gpu.py) never reach the regime bench.py runs in: a persistent 256x256 GEMM grid where every
workgroup walks several work items with bias / GELU / accumulate / column-sum epilogues, flash attention with 384 (batch, head)
pairs, one head per XCD, S = 1024 and PAD-tail tile skipping, the two-stream backward at T = 32768. Here each of those is compared
with an fp64 reference of the same op (full tensors), and the whole bf16 step with the exact-f32 instantiation of the same engine,
which is itself gated against the reference's golden vectors (test_model_gpu.py::test_g10_cfg2_shape_spot_check).

Tolerances are printed with every assert; bf16 output rounding is 2^-9 relative per element, so "rel" = max|a-b| / max|b|."""
import math
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLD, load_vocab, randomize_params, sd_checksum, synth_octuple_batch

pytestmark = pytest.mark.gpu
E2W, W2E = load_vocab()
T_BENCH, D, FF, H = 32768, 768, 3072, 12
BF = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from pianobart_amd import ops as o
    return o


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _mk(g, *shape, scale=1.0):
    return (torch.randn(*shape, device='cuda', generator=g) * scale).to(BF)


# ------------------------------------------------------------------------------------------------ (i) GEMMs, default dispatch
@pytest.mark.parametrize('grid', ['persistent', 'plain'])
def test_fc1_bias_gelu_aux_at_bench_shape(ops, grid):
    """NT 32768 x 3072 x 768 + bias + GELU (+ derivative out): 1536 tiles = 6 work items per persistent workgroup (grid 'plain' =
    the ordinary-grid launch the data-parallel backward uses)."""
    g = torch.Generator(device='cuda').manual_seed(1)
    x, w, b = _mk(g, T_BENCH, D), _mk(g, FF, D, scale=1 / math.sqrt(D)), torch.randn(FF, device='cuda', generator=g)
    out = torch.full((T_BENCH, FF), float('nan'), device='cuda', dtype=BF); aux = torch.full_like(out, float('nan'))
    ops.gemm(x, w, out, M=T_BENCH, N=FF, K=D, dtype=ops.PB_BF16, bias=b, gelu_aux_out=aux, dbg=4096 if grid == 'plain' else 0)
    pre = x.double() @ w.double().t() + b.double()
    ref = torch.nn.functional.gelu(pre)
    r = _rel(out, ref)
    dref = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    rd = _rel(aux, dref)
    print('fc1+gelu rel %.2e  gelu\' rel %.2e (%s grid)' % (r, rd, grid))
    assert r < 6e-3 and rd < 6e-3


@pytest.mark.parametrize('N,K,bias,accum', [(D, FF, True, False), (3 * D, D, True, False), (D, D, True, False), (2 * D, D, True, False),
                                            (D, FF, False, True)])
def test_nt_projections_at_bench_shape(ops, N, K, bias, accum):
    """fc2 / qkv / out-proj / cross-kv projections (NT, + bias) and an accumulating variant: N = 768 is 384 tiles on 256 CUs."""
    g = torch.Generator(device='cuda').manual_seed(N + K)
    x, w = _mk(g, T_BENCH, K), _mk(g, N, K, scale=1 / math.sqrt(K))
    b = torch.randn(N, device='cuda', generator=g) if bias else None
    c0 = _mk(g, T_BENCH, N)
    out = c0.clone() if accum else torch.full((T_BENCH, N), float('nan'), device='cuda', dtype=BF)
    ops.gemm(x, w, out, M=T_BENCH, N=N, K=K, dtype=ops.PB_BF16, bias=b, accum=accum)
    ref = x.double() @ w.double().t()
    if bias:
        ref += b.double()
    if accum:
        ref += c0.double()
    r = _rel(out, ref)
    print('NT N=%d K=%d bias=%s accum=%s rel %.2e' % (N, K, bias, accum, r))
    assert r < 6e-3


def test_logits_gemm_f32_out_at_bench_shape(ops):
    """The 8 LM heads as one 32768 x 1280 x 768 GEMM with f32 logits + bias (model.py:119-126)."""
    g = torch.Generator(device='cuda').manual_seed(5)
    x, w, b = _mk(g, T_BENCH, D), _mk(g, 1280, D, scale=1 / math.sqrt(D)), torch.randn(1280, device='cuda', generator=g)
    out = torch.full((T_BENCH, 1280), float('nan'), device='cuda')
    ops.gemm(x, w, out, M=T_BENCH, N=1280, K=D, dtype=ops.PB_BF16, bias=b, c_f32=True)
    r = _rel(out, x.double() @ w.double().t() + b.double())
    print('logits rel %.2e' % r)
    assert r < 2e-5


@pytest.mark.parametrize('grid', ['persistent', 'plain'])
def test_dgrad_gelu_grad_colsum_at_bench_shape(ops, grid):
    """dU = (dG W2) * gelu'(U) with db1 = column sums of dU from the epilogue registers: NN 32768 x 3072 x 768 (W stored [K][N])."""
    from pianobart_amd._lib import LIB
    g = torch.Generator(device='cuda').manual_seed(7)
    dy, w2, aux = _mk(g, T_BENCH, D), _mk(g, D, FF, scale=1 / math.sqrt(D)), _mk(g, T_BENCH, FF)
    du = torch.full((T_BENCH, FF), float('nan'), device='cuda', dtype=BF)
    cs = torch.full((FF,), 0.5, device='cuda')
    ws = torch.empty(int(LIB.query('pb_gemm_colsum_ws_floats', T_BENCH, FF)), device='cuda')
    ops.gemm(dy, w2, du, M=T_BENCH, N=FF, K=D, dtype=ops.PB_BF16, b_kc=False, lda=D, ldb=FF, ldc=FF, gelu_grad_aux_in=aux, ldaux=FF,
             colsum_out=cs, colsum_ws=ws, dbg=4096 if grid == 'plain' else 0)
    ref = (dy.double() @ w2.double()) * aux.double()
    r, rc = _rel(du, ref), _rel(cs, 0.5 + ref.sum(0))
    print('dfc2*gelu\' rel %.2e  db1 rel %.2e (%s grid)' % (r, rc, grid))
    assert r < 6e-3 and rc < 2e-3


@pytest.mark.parametrize('N,K', [(D, FF), (D, 3 * D), (D, D), (D, 1280)])
def test_dgrad_accumulate_at_bench_shape(ops, N, K):
    """dX += dY W (NN, accumulate into bf16): the fc1 / qkv / out-proj / head input gradients."""
    g = torch.Generator(device='cuda').manual_seed(N * 3 + K)
    dy, w, c0 = _mk(g, T_BENCH, K), _mk(g, K, N, scale=1 / math.sqrt(K)), _mk(g, T_BENCH, N)
    out = c0.clone()
    ops.gemm(dy, w, out, M=T_BENCH, N=N, K=K, dtype=ops.PB_BF16, b_kc=False, lda=K, ldb=N, ldc=N, accum=True)
    r = _rel(out, c0.double() + dy.double() @ w.double())
    print('NN accumulate N=%d K=%d rel %.2e' % (N, K, r))
    assert r < 6e-3


@pytest.mark.parametrize('M,N', [(FF, D), (D, FF), (3 * D, D), (D, D), (1280, D), (2 * D, D)])
def test_wgrad_splitk_at_bench_shape(ops, M, N):
    """G (M,N) f32 = dY(T,M)^T X(T,N), K = T = 32768, split-K into f32 slabs exactly as Engine._wgrad chooses it."""
    g = torch.Generator(device='cuda').manual_seed(M + 7 * N)
    dy, x = _mk(g, T_BENCH, M), _mk(g, T_BENCH, N)
    big = M >= 256 and N >= 256 and M * N > 768 * 768
    tl = 256 if big else 128
    tiles = ((M + tl - 1) // tl) * ((N + tl - 1) // tl)
    nsplit = max(1, min(32, T_BENCH // 64, round((192 if big else 512) / tiles)))
    slabs = torch.empty(nsplit * M * N, device='cuda')
    G = torch.full((M, N), float('nan'), device='cuda')
    ops.gemm(dy, x, G, M=M, N=N, K=T_BENCH, dtype=ops.PB_BF16, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N, c_f32=True, splitk=nsplit,
             slabs=slabs, tile256=big)
    r = _rel(G, dy.double().t() @ x.double())
    print('TN wgrad %dx%d split %d rel %.2e' % (M, N, nsplit, r))
    assert r < 2e-5


# ------------------------------------------------------------------------------------------------ (ii) flash attention
def _attn_ref(qkv, km, causal, b, scale, dout):
    """fp64 reference of one batch row, all heads: returns out (S,d), lse (H,S), dqkv (S,3d)."""
    S, d3 = qkv.shape[1:]
    d = d3 // 3
    hd = d // H
    x = qkv[b].double().requires_grad_(True)
    q = x[:, :d].reshape(S, H, hd).permute(1, 0, 2); k = x[:, d:2 * d].reshape(S, H, hd).permute(1, 0, 2)
    v = x[:, 2 * d:].reshape(S, H, hd).permute(1, 0, 2)
    vis = (km[b] != 0)[None, None, :].expand(H, S, S)
    if causal:
        vis = vis & torch.ones(S, S, dtype=torch.bool, device=qkv.device).tril()
    s = (q @ k.transpose(1, 2) * scale).masked_fill(~vis, float('-inf'))
    p = torch.nan_to_num(torch.softmax(s, -1), nan=0.0)
    o = (p @ v).permute(1, 0, 2).reshape(S, d)
    o.backward(dout[b].double())
    lse = torch.logsumexp(s, -1)
    return o.detach(), lse.detach(), x.grad, vis.any(-1)


@pytest.mark.parametrize('causal,tail', [(False, 'pad'), (True, 'pad'), (False, 'scattered')])
def test_flash_attention_at_bench_shape(ops, causal, tail):
    """pb_flash_fwd / pb_flash_bwd at B = 32, H = 12, S = 1024, head_dim 64 (384 (batch, head) pairs, one head per XCD, kmax tile
    skipping) against an fp64 reference of EVERY (batch, head): outputs, lse, dQ/dK/dV and the fused q/k/v bias gradients.
    'pad' = clean PAD tails of length ~U{0..S/2} (decoder / cross masks), 'scattered' = the corrupted encoder input's mask
    (visible MASK rows inside the tail)."""
    from pianobart_amd._lib import LIB
    B, S, hd = 32, 1024, 64
    d = H * hd
    g = torch.Generator(device='cuda').manual_seed(11 + causal)
    qkv = (torch.randn(B, S, 3 * d, device='cuda', generator=g) * 1.5).to(BF)
    L = torch.randint(S // 2, S + 1, (B,), device='cuda', generator=g)
    L[0], L[1], L[2] = S, S // 2, 1                        # full row, shortest bench row, a single visible key
    km = (torch.arange(S, device='cuda')[None, :] < L[:, None]).float()
    if tail == 'scattered':
        km = torch.maximum(km, (torch.rand(B, S, device='cuda', generator=g) < 0.02).float())
        km[3] = 0                                          # a sample with no visible key at all: zero rows
    kmax = torch.empty(B, dtype=torch.int32, device='cuda')
    ops.key_extent(km, kmax)
    out = torch.full((B, S, d), float('nan'), device='cuda', dtype=BF)
    lse = torch.empty(B, H, S, device='cuda')
    scale = hd ** -0.5
    sl = lambda t, off: (t, off, 3 * d, S * 3 * d)
    ops.flash_fwd(sl(qkv, 0), sl(qkv, d), sl(qkv, 2 * d), (out, 0, d, S * d), lse, km, B, H, S, S, hd, scale, causal, kmax=kmax)
    dout = torch.randn(B, S, d, device='cuda', generator=g).to(BF)
    dqkv = torch.full((B, S, 3 * d), float('nan'), device='cuda', dtype=BF)
    delta = torch.empty(B, H, S, device='cuda')
    db = [torch.full((d,), 0.25, device='cuda') for _ in range(3)]
    dbws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, S, S, hd)), device='cuda')
    ops.flash_bwd(sl(qkv, 0), sl(qkv, d), sl(qkv, 2 * d), (out, 0, d, S * d), dout, lse, km, sl(dqkv, 0), sl(dqkv, d), sl(dqkv, 2 * d), delta,
                  B, H, S, S, hd, scale, causal, kmax=kmax, dbias=db, dbias_ws=dbws)
    torch.cuda.synchronize()
    eo = el = eg = 0.0
    gmax = 0.0
    cs = torch.zeros(3 * d, device='cuda', dtype=torch.double)
    errs = []
    for b in range(B):
        o, l, gq, has = _attn_ref(qkv, km, causal, b, scale, dout)
        eo = max(eo, float((out[b].double() - o).abs().max()))
        if has.any():
            el = max(el, float((lse[b].double() - l)[has].abs().max()))
        errs.append(float((dqkv[b].double() - gq).abs().max()))
        gmax = max(gmax, float(gq.abs().max()))
        cs += gq.sum(0)
    eg = max(errs) / gmax
    eb = max(float((db[i].double() - 0.25 - cs[i * d:(i + 1) * d]).abs().max() / cs.abs().max()) for i in range(3))
    print('flash S=1024 B*H=384 causal=%s %s: out abs %.2e  lse abs %.2e  dqkv rel %.2e  dbias rel %.2e' % (causal, tail, eo, el, eg, eb))
    assert not torch.isnan(out).any() and not torch.isnan(dqkv).any()
    assert eo < 4e-2 and el < 3e-2 and eg < 2e-2 and eb < 1e-2


# ------------------------------------------------------------------------------------------------ (iii) / (iv) whole step
def _pair(S, d, L, f, h, seed, dropout):
    """The same seeded weights in the exact-f32 and the bf16 instantiation of the engine."""
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    cfg = lambda: BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=L, decoder_layers=L, encoder_ffn_dim=f, decoder_ffn_dim=f,
                             encoder_attention_heads=h, decoder_attention_heads=h, dropout=dropout)
    m32 = PianoBartLM(PianoBart(cfg(), E2W, W2E, precision='fp32'))
    randomize_params(m32, seed)
    mbf = PianoBartLM(PianoBart(cfg(), E2W, W2E, precision='bf16'))
    mbf.load_state_dict(m32.state_dict(), strict=True)
    return m32.train().cuda(), mbf.train().cuda()


def _step(ops, m, batch, seed):
    enc, dec, loss_mask, emask, dmask, target = batch
    eng = m._get_engine()
    eng.bind(enc.device)
    eng._seed = seed
    sums = eng.loss_and_grads(ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask,
                              train=True).double().cpu()
    torch.cuda.synchronize()
    w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
    loss = float(((sums[0:8] / sums[8:16]) * w).sum() / w.sum())
    return eng, loss, sums


def _compare_steps(ops, S, d, L, f, h, B, tol_loss, tol_gn, tol_slot, dropout=0.1):
    m32, mbf = _pair(S, d, L, f, h, 41, dropout)
    batch = [t.cuda() for t in synth_octuple_batch(B, S, seed=1234)]
    e32, l32, s32 = _step(ops, m32, batch, 99)
    g32 = e32.G32.double().clone()
    slots = e32.slots
    del m32, e32
    torch.cuda.empty_cache()
    ebf, lbf, sbf = _step(ops, mbf, batch, 99)
    gbf = ebf.G32.double()
    gn32, gnbf = float(g32.norm()), float(gbf.norm())
    worst, cos_min = ('', 0.0), 1.0
    for name, sl in slots.items():
        a, b = gbf[sl.off:sl.off + sl.numel], g32[sl.off:sl.off + sl.numel]
        nb = float(b.norm())
        if nb < 1e-3 * gn32 / math.sqrt(len(slots)):           # mathematically ~0 gradients (k-proj bias: softmax shift invariance)
            continue
        e = float((a - b).norm()) / nb
        cos_min = min(cos_min, float((a * b).sum() / (a.norm() * b.norm())))
        if e > worst[1]:
            worst = (name, e)
    acc32, accbf = float(s32[16:24].sum() / s32[8:16].sum()), float(sbf[16:24].sum() / sbf[8:16].sum())
    print('%dL/%d S=%d B=%d dropout %.1f: loss fp32 %.6f bf16 %.6f (rel %.2e); grad norm %.5f / %.5f (rel %.2e); worst slot %s %.3f; min cosine %.4f; '
          'masked acc %.4f / %.4f' % (L, d, S, B, dropout, l32, lbf, abs(lbf - l32) / l32, gn32, gnbf, abs(gnbf - gn32) / gn32, worst[0], worst[1],
                                     cos_min, acc32, accbf))
    assert abs(lbf - l32) / l32 < tol_loss
    assert abs(gnbf - gn32) / gn32 < tol_gn
    assert worst[1] < tol_slot, worst
    return ebf, mbf, batch


def test_bf16_step_matches_fp32_step_cfg2_shape(ops):
    """configs[1] model and sequence length, B = 8 (T = 8192, every GEMM on the 256x256 persistent kernel, dropout on with the same
    Philox masks in both precisions): loss, global gradient norm and every parameter tensor's gradient, bf16 vs exact f32. Then the
    bf16 step again on one stream: bit-identical to the two-stream schedule at this size."""
    from pianobart_amd import engine as E
    # bounds ~2x the measured differences (loss 2.4e-5 .. 3.4e-5, gradient norm 3.3e-3 .. 3.7e-3, worst tensor 3.5e-2)
    ebf, mbf, batch = _compare_steps(ops, 1024, 768, 12, 3072, 12, 8, tol_loss=2e-4, tol_gn=1e-2, tol_slot=0.07)
    Te, Td, T, Ts = ebf.last_rows
    assert Te < T and Td < T and Ts < Td, ebf.last_rows      # the bf16 side of this comparison is the PACKED step
    if ebf._side_stream() is not None:
        two = ebf.G32.clone()
        saved = E._WGRAD_STREAM
        try:
            E._WGRAD_STREAM = 0
            _step(ops, mbf, batch, 99)
        finally:
            E._WGRAD_STREAM = saved
        assert torch.equal(two, ebf.G32), float((two - ebf.G32).abs().max())


def test_bf16_step_at_full_bench_batch_is_finite_and_consistent(ops):
    """B = 32, S = 1024 (exactly the bench step): the bf16 loss equals the B = 8 sub-batches' mask-weighted mean computed by the same
    engine (dropout off), i.e. the T = 32768 launch configuration computes what the T = 8192 one does."""
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    cfg = BartConfig(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072,
                     decoder_ffn_dim=3072, encoder_attention_heads=12, decoder_attention_heads=12, dropout=0.0)
    m = PianoBartLM(PianoBart(cfg, E2W, W2E, precision='bf16'))
    randomize_params(m, 41)
    m = m.train().cuda()
    batch = [t.cuda() for t in synth_octuple_batch(32, 1024, seed=1234)]
    eng, loss, sums = _step(ops, m, batch, 5)
    assert math.isfinite(loss) and bool(torch.isfinite(eng.G32).all())
    Te, Td, T, Ts = eng.last_rows
    assert T == 32768 and Te < T and Td < T and Ts < Td, eng.last_rows      # the bench's own row counts: packed, last layer on the loss rows
    g_full = eng.G32.double().clone()
    parts = torch.zeros(24, dtype=torch.double)
    for i in range(4):
        _, _, s = _step(ops, m, [t[8 * i:8 * i + 8].contiguous() for t in batch], 5)
        parts += s
    print('B=32 sums vs 4 x B=8: max rel diff %.2e' % float(((parts - sums).abs() / sums.abs().clamp_min(1)).max()))
    assert torch.allclose(parts[8:16], sums[8:16]) and torch.allclose(parts[16:24], sums[16:24], atol=6.0)      # counts exact; a few argmax near-ties may flip (measured: <= 4 of 4952 rows)
    assert float(((parts[0:8] - sums[0:8]).abs() / sums[0:8]).max()) < 2e-3
    assert float(g_full.norm()) > 0


def test_bf16_step_matches_fp32_step_cfg5_shape(ops):
    """configs[4] shape: 24L / 1024 / ffn 4096 / 16 heads, S = 2048, B = 1 (d = 1024 row kernels, head_dim 64 at S = 2048)."""
    _compare_steps(ops, 2048, 1024, 24, 4096, 16, 1, tol_loss=5e-4, tol_gn=1e-2, tol_slot=0.1)      # measured 1.1e-4 / 2.8e-3 / 4.7e-2


def test_g10_cfg2_shape_spot_check_bf16():
    """The reference's own cfg-2-shape vectors (G10: 12L/768/ffn3072, S = 1024, B = 1) against the bf16 instantiation: reported
    tolerance (not the 1e-3 of the exact-f32 instantiation) and argmax agreement."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    z = np.load(os.path.join(GOLD, 'g10_cfg2_spot.npz'))
    cfg = BartConfig(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072,
                     decoder_ffn_dim=3072, encoder_attention_heads=12, decoder_attention_heads=12)
    m = PianoBartLM(PianoBart(cfg, E2W, W2E, precision='bf16'))
    randomize_params(m, 41)
    m = m.eval()
    assert sd_checksum(m.state_dict()) == bytes(z['sd_sha']).decode()
    m = m.cuda()
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(1, 1024, seed=19)]
    with torch.no_grad():
        y = torch.cat(m(enc, dec, emask, dmask), dim=-1)[0].float().cpu()
    rows = z['rows']
    rel = float((y[rows] - torch.from_numpy(z['logit_rows'])).abs().max() / float(z['logit_absmax']))
    offs = np.cumsum([0, 262, 134, 135, 262, 134, 38, 260, 55])
    arg = torch.stack([y[:, offs[i]:offs[i + 1]].argmax(-1) for i in range(8)], dim=-1).numpy()
    ref = z['argmax'].astype(np.int64)
    agree = float((arg == ref).mean())
    clear = z['top2_gap'] > 5e-2 * float(z['logit_absmax'])
    print('cfg2 bf16 logits rel = %.3e; argmax agreement %.4f overall, %.4f of the %.1f %% with a top-2 gap > 5%% of max|logit|'
          % (rel, agree, float((arg[clear] == ref[clear]).mean()), 100 * clear.mean()))
    assert rel < 2.5e-2                                  # measured 1.4e-2 - 1.5e-2
    assert agree > 0.995 and np.array_equal(arg[clear], ref[clear])


def test_bf16_loss_curve_follows_the_exact_f32_curve():
    """SURVEY 7 hard part 1 / VERDICT r4 item 6: the bf16 throughput instantiation against the exact-f32 parity instantiation as TRAINING runs --
    cfg-2 model (12L / 768d / S = 1024), same initial weights, same batches in the same order, same Philox dropout seeds, fused step + HF-AdamW
    at lr 1e-4 (tools/loss_overlay.py; 200 steps at B = 8 are committed as profiles/r05_loss_overlay.txt: gap <= 6e-4 for the first 50 steps,
    then the two trajectories decorrelate to 1 - 3.5 %, final losses 1.815 / 1.791). Here, reduced: B = 2, 24 steps."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import types
    from tools import loss_overlay as LO
    args = types.SimpleNamespace(steps=24, batch=2, nbatch=4, seq=1024, layers=12, hs=768, ffn=3072, heads=12, dropout=0.1, lr=1e-4)
    lb, lf = LO.overlay(args)
    gap = [abs(a - b) / b for a, b in zip(lb, lf)]
    print('bf16 vs f32 loss, 24 steps: first %.5f / %.5f, last %.5f / %.5f, max gap %.2e' % (lb[0], lf[0], lb[-1], lf[-1], max(gap)))
    assert gap[0] < 1e-4                                                  # the same forward at step 0 (measured 2e-6 at B = 8)
    assert max(gap) < 3e-2 and sum(gap[-4:]) / 4 < 2e-2                   # measured at B = 2: max 1.07e-2 (the B = 8 / 200-step run peaks at 3.5e-2)
    assert lf[-1] < 0.8 * lf[0] and lb[-1] < 0.8 * lb[0]                  # both learn


def test_split_bf16_loss_curve_stays_on_the_exact_f32_curve():
    """The bf16x3 parity instantiation as a TRAINING run against the exact-f32 one (same weights, batches, Philox dropout bits; tools/loss_overlay.py
    --precision bf16x3; 200 steps at B = 8 are committed as profiles/r06_loss_overlay_bf16x3.txt: gap <= 2.3e-5 for 50 steps, 9e-4 to step 150). Here: B = 2, 24 steps."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import types
    from tools import loss_overlay as LO
    args = types.SimpleNamespace(steps=24, batch=2, nbatch=4, seq=1024, layers=12, hs=768, ffn=3072, heads=12, dropout=0.1, lr=1e-4, precision='bf16x3')
    lx, lf = LO.overlay(args)
    gap = [abs(a - b) / b for a, b in zip(lx, lf)]
    print('bf16x3 vs f32 loss, 24 steps: first %.6f / %.6f, last %.6f / %.6f, max gap %.2e' % (lx[0], lf[0], lx[-1], lf[-1], max(gap)))
    assert gap[0] < 1e-6 and max(gap) < 1e-4
    assert lf[-1] < 0.8 * lf[0]
