"""Dead-row compaction (packed rows): the row-map kernels against numpy, and the packed attention / embedding kernels against the
dense kernels on the same rows. A packed batch holds, for every sequence, its visible rows first, so on those rows the packed
kernels walk exactly the tiles the dense ones walk: outputs agree bit for bit (but for the wave-wide lazy-maximum decision at the end
of a sequence, see `same`); column sums (bias gradients) are summed in a different block structure and agree to f32 rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from pianobart_amd import ops as o
    return o


def _i32(x):
    return torch.tensor(np.asarray(x), dtype=torch.int32, device='cuda')


def _rowmap_numpy(mask, loss_mask, off, length):
    B, S = mask.shape
    row_src = -np.ones(int(off[-1] + length[-1]), np.int64)
    inv = -np.ones((B, S), np.int64)
    for b in range(B):
        vis = [s for s in range(S) if mask[b, s] != 0]
        los = [s for s in range(S) if mask[b, s] == 0 and loss_mask is not None and loss_mask[b, s].any()]
        dead = [s for s in range(S) if s not in set(vis) | set(los)]
        order = (vis + los + dead)[:length[b]]
        for i, s in enumerate(order):
            row_src[off[b] + i] = b * S + s
            inv[b, s] = off[b] + i
    return row_src, inv


def test_rowmap_kernels_match_numpy(ops):
    rng = np.random.default_rng(0)
    B, S = 5, 333
    emask = (rng.random((B, S)) > 0.3).astype(np.float32)
    L = rng.integers(1, S, size=B)
    dmask = (np.arange(S)[None, :] < L[:, None]).astype(np.float32)
    dmask[3, 5] = 0                                                      # batch 3: not a prefix mask
    lm = np.zeros((B, S, 8), np.float32)
    lm[rng.random((B, S)) > 0.8, 3] = 1.0
    counts = torch.empty(B, 8, dtype=torch.int32, device='cuda')
    te, td, tl = (torch.tensor(x, device='cuda') for x in (emask, dmask, lm))
    ops.rowmap_count(te, td, tl, counts)
    c = counts.cpu().numpy()
    assert (c[:, 0] == (emask != 0).sum(1)).all() and (c[:, 1] == (dmask != 0).sum(1)).all()
    assert (c[:, 2] == ((dmask != 0) | lm.any(-1)).sum(1)).all()
    assert c[:, 3].tolist() == [1, 1, 1, 0, 1] and (c[:, 4] == lm.any(-1).sum(1)).all()
    for mask, t_mask, loss, t_loss in ((emask, te, None, None), (dmask, td, lm, tl)):
        live = (mask != 0).sum(1) if loss is None else ((mask != 0) | loss.any(-1)).sum(1)
        length = np.minimum(live + rng.integers(0, 7, size=B), S)       # a few dead fillers per sequence
        off = np.concatenate([[0], np.cumsum(length)[:-1]])
        Tp = int(length.sum())
        row_src, row_pos, inv = (torch.full((n,), -7, dtype=torch.int32, device='cuda') for n in (Tp, Tp, B * S))
        ops.rowmap_build(t_mask, t_loss, _i32(off), _i32(length), row_src, row_pos, inv)
        want_src, want_inv = _rowmap_numpy(mask, loss, off, length)
        assert (row_src.cpu().numpy() == want_src).all()
        assert (row_pos.cpu().numpy() == want_src % S).all()
        assert (inv.cpu().numpy().reshape(B, S) == want_inv).all()
        # gather + the position-table sum through the inverse map
        src = torch.randn(B * S, 8, device='cuda')
        dst = torch.empty(Tp, 8, device='cuda')
        ops.gather_rows16(src, row_src, dst, Tp, 32)
        assert torch.equal(dst, src[row_src.long()])
        x = torch.randn(Tp, 64, device='cuda').to(torch.bfloat16)
        out = torch.ones(S, 64, device='cuda')
        ops.pos_grad_packed(x, inv, out, B, S)
        want = torch.ones(S, 64, device='cuda', dtype=torch.float64)
        want.index_add_(0, (row_src % S).long(), x.double())
        assert float((out.double() - want).abs().max()) < 1e-5
    # the loss rows of the decoder packing (+ a few other rows of it): what the last decoder layer's query side runs on
    inv_np = inv.cpu().numpy().reshape(B, S)
    nloss = lm.any(-1).sum(1)
    len_s = np.minimum(nloss + 3, length)
    off_s = np.concatenate([[0], np.cumsum(len_s)[:-1]])
    Ts = int(len_s.sum())
    src_s, idx_s = (torch.full((Ts,), -7, dtype=torch.int32, device='cuda') for _ in range(2))
    ops.rowmap_build_sub(tl, inv, _i32(off_s), _i32(len_s), src_s, idx_s)
    got_src, got_idx = src_s.cpu().numpy(), idx_s.cpu().numpy()
    for b in range(B):
        los = [s for s in range(S) if lm[b, s].any()]
        oth = [s for s in range(S) if not lm[b, s].any() and inv_np[b, s] >= 0]
        order = (los + oth)[:len_s[b]]
        assert got_src[off_s[b]:off_s[b] + len_s[b]].tolist() == [b * S + s for s in order]
        assert got_idx[off_s[b]:off_s[b] + len_s[b]].tolist() == [inv_np[b, s] for s in order]
    # scatter is the inverse of gather
    full = torch.randn(Tp, 8, device='cuda')
    part = torch.empty(Ts, 8, device='cuda')
    ops.gather_rows16(full, idx_s, part, Ts, 32)
    back = torch.zeros_like(full)
    ops.scatter_rows16(part, idx_s, back, Ts, 32)
    keep = torch.zeros(Tp, dtype=torch.bool, device='cuda'); keep[idx_s.long()] = True
    assert torch.equal(back[keep], full[keep]) and not back[~keep].any()


@pytest.mark.parametrize('hd', [64, 96, 128])
@pytest.mark.parametrize('kind', ['self', 'causal', 'cross'])
def test_packed_attention_equals_dense_on_the_kept_rows(ops, hd, kind):
    g = torch.Generator(device='cuda').manual_seed(hd)
    B, H, S = 4, 3, 384
    d = H * hd
    scale = hd ** -0.5
    causal = kind == 'causal'
    kvis = [300, 129, 384, 64]                                         # visible keys (a prefix of each sequence)
    klen = [310, 129, 384, 200]                                        # key rows kept (the rest of the kept rows are invisible)
    qlen = klen if kind != 'cross' else [257, 384, 100, 128]
    q = (torch.randn(B, S, d, device='cuda', generator=g) * 1.5).to(torch.bfloat16)
    kv = (torch.randn(B, S, 2 * d, device='cuda', generator=g) * 1.5).to(torch.bfloat16)
    dout = torch.randn(B, S, d, device='cuda', generator=g).to(torch.bfloat16)
    for b in range(B):
        dout[b, qlen[b]:] = 0                                          # dropped query rows: dead in the dense run too
    km = torch.zeros(B, S, device='cuda')
    for b in range(B):
        km[b, :kvis[b]] = 1
    kmax = _i32(kvis)
    # dense run
    o_d = torch.zeros(B, S, d, device='cuda', dtype=torch.bfloat16)
    lse_d, delta_d = torch.empty(B, H, S, device='cuda'), torch.empty(B, H, S, device='cuda')
    dq_d, dkv_d = torch.zeros_like(q), torch.zeros_like(kv)
    ex = lambda t, off, ld: (t, off, ld, S * ld)
    ops.flash_fwd(ex(q, 0, d), ex(kv, 0, 2 * d), ex(kv, d, 2 * d), ex(o_d, 0, d), lse_d, km, B, H, S, S, hd, scale, causal, kmax=kmax)
    need = int(ops.LIB.query('pb_flash_bias_ws_floats', B, H, S, S, hd))
    wsb = torch.empty(need, device='cuda')
    db_d = [torch.zeros(d, device='cuda') for _ in range(3)]
    ops.flash_bwd(ex(q, 0, d), ex(kv, 0, 2 * d), ex(kv, d, 2 * d), ex(o_d, 0, d), dout, lse_d, km, ex(dq_d, 0, d), ex(dkv_d, 0, 2 * d),
                  ex(dkv_d, d, 2 * d), delta_d, B, H, S, S, hd, scale, causal, kmax=kmax, dbias=db_d, dbias_ws=wsb)
    # packed run on the kept rows
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]])
    koff = np.concatenate([[0], np.cumsum(klen)[:-1]])
    pk = lambda t, lens: torch.cat([t[b, :lens[b]] for b in range(B)]).contiguous()
    qp, kvp, doutp = pk(q, qlen), pk(kv, klen), pk(dout, qlen)
    Tq, Tk = qp.shape[0], kvp.shape[0]
    rows = ops.PackedRows(_i32(qoff), _i32(qlen), _i32(koff), _i32(klen), kmax, max(qlen), max(klen))
    o_p = torch.full((Tq, d), float('nan'), device='cuda', dtype=torch.bfloat16)
    lse_p = torch.full((B, H, max(qlen)), float('nan'), device='cuda')
    delta_p = torch.empty(B, H, max(qlen), device='cuda')
    dq_p = torch.full((Tq, d), float('nan'), device='cuda', dtype=torch.bfloat16)
    dkv_p = torch.full((Tk, 2 * d), float('nan'), device='cuda', dtype=torch.bfloat16)
    ops.flash_fwd_packed((qp, 0, d), (kvp, 0, 2 * d), (kvp, d, 2 * d), (o_p, 0, d), lse_p, rows, B, H, hd, scale, causal)
    db_p = [torch.zeros(d, device='cuda') for _ in range(3)]
    wsb2 = torch.full((int(ops.LIB.query('pb_flash_bias_ws_floats', B, H, max(qlen), max(klen), hd)),), float('nan'), device='cuda')
    ops.flash_bwd_packed((qp, 0, d), (kvp, 0, 2 * d), (kvp, d, 2 * d), (o_p, 0, d), doutp, lse_p, (dq_p, 0, d), (dkv_p, 0, 2 * d),
                         (dkv_p, d, 2 * d), delta_p, rows, B, H, hd, scale, causal, dbias=db_p, dbias_ws=wsb2)
    torch.cuda.synchronize()

    def same(a, b_):
        """Bit-identical, except where a wave of the dense run took the exact softmax path because of a row the packed run does
        not have (the lazy-maximum decision is per wave, i.e. per 32 query rows): there the two runs round differently, by one bf16
        ulp, in a few rows of a sequence's last row block."""
        a, b_ = a.float(), b_.float()
        diff = (a - b_).abs()
        assert float(diff.max()) <= 2.0 ** -6 * max(1.0, float(b_.abs().max())), float(diff.max())
        assert float((diff > 0).float().mean()) < 0.01, float((diff > 0).float().mean())

    same(o_p, pk(o_d, qlen))
    for b in range(B):
        same(lse_p[b, :, :qlen[b]], lse_d[b, :, :qlen[b]])
    same(dq_p, pk(dq_d, qlen))
    same(dkv_p, pk(dkv_d, klen))
    for b in range(B):                                                 # invisible kept key rows get exact zeros
        assert not dkv_p[koff[b] + kvis[b]:koff[b] + klen[b]].any()
    for a, b_ in zip(db_p, db_d):
        assert torch.isfinite(a).all()
        assert float((a - b_).abs().max()) <= 5e-3 * max(1.0, float(b_.abs().max()))     # f32 sums in another order (+ the rows `same` describes)


def test_packed_embedding_equals_dense_rows(ops):
    from tests.golden_util import synth_octuple_batch
    B, S, d = 3, 128, 256
    enc = synth_octuple_batch(B, S, seed=5)[0].cuda()
    ids16 = ops.ids_to_i16(enc)
    g = torch.Generator(device='cuda').manual_seed(1)
    P = torch.randn(ops.TAB_TOTAL, d, device='cuda', generator=g)
    lin_b, pos = torch.randn(d, device='cuda', generator=g), torch.randn(S + 2, d, device='cuda', generator=g)
    w, bb = torch.randn(d, device='cuda', generator=g), torch.randn(d, device='cuda', generator=g)
    y = torch.empty(B * S, d, device='cuda', dtype=torch.bfloat16)
    m, r = torch.empty(B * S, device='cuda'), torch.empty(B * S, device='cuda')
    seed, pd = 1234, 0.1                                               # dropout on: a packed row draws the bits of its row in the padded batch
    ops.embed_ln_fwd(ids16, P, lin_b, pos, w, bb, y, m, r, S, 1e-5, seed, 0, pd, padded=True)
    keep = torch.tensor([5, 0, 130, 131, 255, 300, 383, 17], device='cuda', dtype=torch.int32)
    ids_p = torch.empty(len(keep), 8, dtype=torch.int16, device='cuda')
    ops.gather_rows16(ids16, keep, ids_p, len(keep), 16)
    yp = torch.empty(len(keep), d, device='cuda', dtype=torch.bfloat16)
    mp, rp = torch.empty(len(keep), device='cuda'), torch.empty(len(keep), device='cuda')
    ops.embed_ln_fwd(ids_p, P, lin_b, pos, w, bb, yp, mp, rp, S, 1e-5, seed, 0, pd, padded=True, row_ids=keep)
    assert torch.equal(yp, y[keep.long()]) and torch.equal(mp, m[keep.long()]) and torch.equal(rp, r[keep.long()])
    # backward (dz route): dz rows equal the dense dz rows
    partials = torch.empty(int(ops.LIB.query('pb_ln_partials_floats', d)), device='cuda')
    dy = torch.randn(B * S, d, device='cuda', generator=g).to(torch.bfloat16)
    dz = torch.empty_like(dy)
    gv = lambda: [torch.zeros(d, device='cuda') for _ in range(3)]
    ga = gv()
    ops.embed_ln_bwd(dy, ids16, P, lin_b, pos, w, m, r, None, None, ga[0], ga[1], ga[2], partials, S, seed, 0, pd, dz_out=dz, padded=True)
    dzp = torch.empty(len(keep), d, device='cuda', dtype=torch.bfloat16)
    gb = gv()
    ops.embed_ln_bwd(dy[keep.long()].contiguous(), ids_p, P, lin_b, pos, w, mp, rp, None, None, gb[0], gb[1], gb[2], partials, S, seed, 0, pd,
                     dz_out=dzp, padded=True, row_ids=keep)
    assert torch.equal(dzp, dz[keep.long()])
    # residual + dropout + LayerNorm on packed rows: the same bits, so the same rows
    res = torch.randn(B * S, d, device='cuda', generator=g).to(torch.bfloat16)
    a = torch.randn(B * S, d, device='cuda', generator=g).to(torch.bfloat16)
    ops.add_ln_fwd(res, a, w, bb, y, m, r, 1e-5, seed, 5, pd)
    kl = keep.long()
    ops.add_ln_fwd(res[kl].contiguous(), a[kl].contiguous(), w, bb, yp, mp, rp, 1e-5, seed, 5, pd, row_ids=keep)
    assert torch.equal(yp, y[kl]) and torch.equal(mp, m[kl]) and torch.equal(rp, r[kl])
    dres, da = torch.empty_like(dy), torch.empty_like(dy)
    ops.add_ln_bwd(dy, res, a, w, m, r, dres, da, ga[0], ga[1], ga[2], partials, False, seed, 5, pd)
    dresp, dap = torch.empty_like(dzp), torch.empty_like(dzp)
    ops.add_ln_bwd(dy[kl].contiguous(), res[kl].contiguous(), a[kl].contiguous(), w, mp, rp, dresp, dap, gb[0], gb[1], gb[2], partials, False, seed, 5, pd,
                   row_ids=keep)
    assert torch.equal(dresp, dres[kl]) and torch.equal(dap, da[kl])


def _step(eng, args, pack, monkeypatch, seed=77, train=True):
    from pianobart_amd import engine as E
    monkeypatch.setattr(E, '_PACK_ROWS', 1 if pack else 0)
    monkeypatch.setattr(E, '_ATTN_BWD1', 0)      # the same backward arithmetic on both sides (the padded step's one-pass form rounds dQ once more:
    eng._seed = seed                             # test_one_pass_backward_in_the_padded_step below holds it against this one)
    sums = eng.loss_and_grads(*args, train=train).clone()
    torch.cuda.synchronize()
    return sums, eng.G32.clone(), eng.last_rows


@pytest.mark.parametrize('heads,tail_loss,hole,dropout,sparse', [(4, False, False, 0.0, False), (4, True, False, 0.1, True), (2, True, True, 0.1, True),
                                                                 (4, False, False, 0.0, True), (-4, True, False, 0.1, True)])
def test_packed_step_equals_dense_step(ops, monkeypatch, heads, tail_loss, hole, dropout, sparse):
    """The fused pre-train step with the dead rows dropped gives the dense step's loss sums and gradients, dropout included (a
    packed row draws the dropout bits of its row in the padded batch). The comparison is to bf16 rounding: the
    GEMMs pick their tile shape by the row count, so a kept row's f32 sums are not accumulated in the same order in both runs
    (measured: hidden states differ by one bf16 ulp, gradient slots by 1e-3 .. 5e-3 relative); the kernels themselves are compared
    bit for bit above."""
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    e2w, w2e = load_vocab()
    B, S, d = 6, 256, 256
    if heads < 0:                                                       # head_dim 96 (the secondary cfg-2 shape's head size)
        heads, d = -heads, 384
    cfg = BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                     encoder_attention_heads=heads, decoder_attention_heads=heads, dropout=dropout)
    m = PianoBartLM(PianoBart(cfg, e2w, w2e, precision='bf16'))
    randomize_params(m, 11)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(B, S, seed=9)]
    rng = np.random.default_rng(3)
    Le, Ld = rng.integers(40, S + 1, size=B), rng.integers(40, S + 1, size=B)
    Le[0], Ld[1] = S, S
    emask, dmask, loss_mask = emask.clone().float(), dmask.clone().float(), loss_mask.clone().float()
    for b in range(B):
        emask[b, :Le[b]] = 1; emask[b, Le[b]:] = 0
        if hole:
            emask[b, 3] = 0                                             # an invisible encoder row inside the sequence
        dmask[b, :Ld[b]] = 1; dmask[b, Ld[b]:] = 0
        loss_mask[b, Ld[b]:] = 0
        loss_mask[b, :Ld[b], 0] = torch.tensor((rng.random(Ld[b]) < 0.15).astype(np.float32), device='cuda') if sparse else 1
        loss_mask[b, 1, 0] = 1
        if tail_loss and Ld[b] + 5 < S:
            loss_mask[b, Ld[b] + 2, :] = 1                              # loss terms on rows that are not visible as keys
            loss_mask[b, S - 1, 2] = 1
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    s0, g0, r0 = _step(eng, args, False, monkeypatch)
    s1, g1, r1 = _step(eng, args, True, monkeypatch)
    assert r0 == (B * S,) * 4 and r1[0] < B * S and r1[1] < B * S and r1[0] % 256 == 0 and r1[1] % 256 == 0, (r0, r1)
    assert (r1[3] < r1[1] and r1[3] % 256 == 0) if sparse else r1[3] == r1[1], r1     # sparse loss rows: the last decoder layer's query side shrinks
    assert torch.equal(s0[8:16], s1[8:16])                              # the mask counts
    rt, gt = 1e-3, (4e-2 if hole else 2e-2)      # a hole in the encoder mask moves the later keys up by one: other tiles, other roundings
    assert torch.allclose(s0[:8], s1[:8], rtol=rt), (s0 - s1).abs().max()
    assert float((s0[16:] - s1[16:]).abs().max()) <= 2                  # argmax hits: a near tie may flip
    assert torch.isfinite(g1).all()
    worst = ('', 0.0)
    for name, sl in eng.slots.items():
        a, b_ = g0[sl.off:sl.off + sl.numel], g1[sl.off:sl.off + sl.numel]
        err = float((a - b_).norm()) / (float(a.norm()) + 1e-12)
        worst = max(worst, (name, err), key=lambda t: t[1])
    print('packed vs dense: worst gradient slot', worst)
    assert worst[1] < gt, worst
    # evaluation (no backward) packs too
    e0 = _step(eng, args, False, monkeypatch, train=False)[0]
    e1 = _step(eng, args, True, monkeypatch, train=False)[0]
    assert torch.allclose(e0[:16], e1[:16], rtol=rt) and float((e0[16:] - e1[16:]).abs().max()) <= 2
    # a decoder mask with a hole cannot be packed: the step stays dense
    dm2 = dmask.clone(); dm2[2, 7] = 0
    r2 = _step(eng, args[:5] + (dm2,), True, monkeypatch)[2]
    assert r2 == (B * S,) * 4


@pytest.mark.parametrize('M,N,K,lay', [(26624, 768, 3072, 'NT'), (26880, 768, 2304, 'NN'), (26624, 768, 768, 'NT'), (32768, 768, 3072, 'NN'),
                                       (3328, 768, 3072, 'NT'), (26624 + 40, 768, 1536, 'NN')])
@pytest.mark.parametrize('epi', ['bias', 'accum', 'f32'])
def test_gemm_tail_split_matches_whole_tiles(ops, M, N, K, lay, epi):
    """With PB_GEMM_TAIL_SPLIT the persistent 256x256 GEMM may cut the tiles of a partly filled last round into K ranges
    (pb_gemm2.hip). Against the same GEMM without the flag: equal up to the order of the f32 partial sums (one bf16 ulp after
    rounding), and both against an fp32 matmul."""
    from pianobart_amd._lib import PB_BF16
    g = torch.Generator(device='cuda').manual_seed(M + K)
    A = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    Bm = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).to(torch.bfloat16)
    Bop = Bm if lay == 'NT' else Bm.t().contiguous()                   # NN: B stored [K][N]
    bias = torch.randn(N, device='cuda', generator=g) if epi == 'bias' else None
    c32 = epi == 'f32'
    C0 = torch.randn(M, N, device='cuda', generator=g).to(torch.float32 if c32 else torch.bfloat16)
    outs = []
    for flags in (0, 32768):
        C = C0.clone()
        ops.gemm(A, Bop, C, M=M, N=N, K=K, dtype=PB_BF16, b_kc=(lay == 'NT'), bias=bias, accum=(epi == 'accum'), c_f32=c32, dbg=flags)
        outs.append(C.float())
    ref = A.float() @ Bm.float().t()
    if bias is not None:
        ref += bias
    if epi == 'accum':
        ref += C0.float()
    tol = 1e-4 if c32 else 2e-2
    for o in outs:
        assert float((o - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    assert float((outs[0] - outs[1]).abs().max()) <= (1e-5 if c32 else 1.6e-2) * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('M,N,K,lay', [(26624, 768, 768, 'NT'), (26880, 768, 2304, 'NT'), (26624, 768, 3072, 'NN'), (26880, 768, 1536, 'NT'),
                                       (26624 + 40, 768, 768, 'NT'), (26624, 1536, 768, 'NT')])
@pytest.mark.parametrize('epi', ['bias', 'accum', 'f32', 'gelu'])
def test_gemm_row_split_matches_one_launch(ops, M, N, K, lay, epi):
    """Round 3: the rows of a partly filled last round of the persistent 256x256 grid go to a second launch of the 128x128 kernel
    (pb_gemm2_try, "Row split", PB_GEMM_ROW_SPLIT = 65536). Against the same GEMM without the flag: every output row written exactly once --
    equal to the order of the f32 sums inside a tile (one bf16 ulp) -- and both against an fp32 matmul; the GELU form also checks
    the derivative rows the split launch writes through its own aux pointer."""
    from pianobart_amd._lib import PB_BF16
    g = torch.Generator(device='cuda').manual_seed(M + K + N)
    A = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    Bm = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).to(torch.bfloat16)
    Bop = Bm if lay == 'NT' else Bm.t().contiguous()
    bias = torch.randn(N, device='cuda', generator=g) if epi in ('bias', 'gelu') else None
    c32 = epi == 'f32'
    C0 = torch.randn(M, N, device='cuda', generator=g).to(torch.float32 if c32 else torch.bfloat16)
    outs, auxs = [], []
    for flags in (0, 65536):
        C = C0.clone()
        aux = torch.full((M, N), 7.0, device='cuda', dtype=torch.bfloat16) if epi == 'gelu' else None
        ops.gemm(A, Bop, C, M=M, N=N, K=K, dtype=PB_BF16, b_kc=(lay == 'NT'), bias=bias, accum=(epi == 'accum'), c_f32=c32, dbg=flags,
                 gelu_aux_out=aux)
        outs.append(C.float()); auxs.append(aux)
    ref = A.float() @ Bm.float().t()
    if bias is not None:
        ref += bias
    if epi == 'accum':
        ref += C0.float()
    if epi == 'gelu':
        u = ref.double()
        cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
        dref = (cdf + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5).float()
        ref = (u * cdf).float()
        for a_ in auxs:
            assert float((a_.float() - dref).abs().max()) < 2e-2
        assert float((auxs[0].float() - auxs[1].float()).abs().max()) <= 1.6e-2
    tol = 1e-4 if c32 else 2e-2
    for o in outs:
        assert float((o - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    assert float((outs[0] - outs[1]).abs().max()) <= (1e-5 if c32 else 1.6e-2) * max(1.0, float(ref.abs().max()))


def test_transpose_batch_bf16(ops):
    """The transposed weight copies the backward's dX = dY W reads (Engine._refresh_wT): several matrices of one flat buffer in one launch."""
    g = torch.Generator(device='cuda').manual_seed(4)
    shapes = [(2304, 768), (768, 768), (8, 16), (3072, 768), (72, 200), (1280, 768)]
    offs, cur = [], 16
    for R, C in shapes:
        offs.append(cur)
        cur += R * C + 24
    src = torch.randn(cur, device='cuda', generator=g).to(torch.bfloat16)
    dst = torch.full_like(src, 7.0)
    table, tiles = [], 0
    for (R, C), o in zip(shapes, offs):
        table.append([o, R, C, tiles])
        tiles += -(-R // 64) * -(-C // 64)
    ops.transpose_batch_bf16(src, dst, torch.tensor(table, dtype=torch.int32, device='cuda'), tiles)
    covered = torch.zeros(cur, dtype=torch.bool, device='cuda')
    for (R, C), o in zip(shapes, offs):
        assert torch.equal(dst[o:o + R * C].view(C, R), src[o:o + R * C].view(R, C).t())
        covered[o:o + R * C] = True
    assert bool((dst[~covered] == 7.0).all())


def test_pipelined_parameter_update_takes_the_same_steps(ops):
    """Engine.pipeline_updates: the AdamW pass and the transposed weight copies run on the second stream beside the next forward,
    which waits group by group. Three training steps must leave bit-identical parameters, moments and loss sums; a state_dict
    taken right after a step must hold the updated values."""
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    e2w, w2e = load_vocab()
    B, S, d = 4, 256, 256
    cfg = BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=3, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                     encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.1)
    batch = [t.cuda() for t in synth_octuple_batch(B, S, seed=21)]
    enc, dec, loss_mask, emask, dmask, target = batch
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    res = []
    for pipelined in (False, True):
        m = PianoBartLM(PianoBart(cfg, e2w, w2e, precision='bf16'))
        randomize_params(m, 11)
        m = m.train().cuda()
        eng = m._get_engine()
        eng.bind(torch.device('cuda', 0))
        eng.pipeline_updates = pipelined
        eng._seed = 5
        sums = []
        for _ in range(3):
            sums.append(eng.loss_and_grads(*args, train=True).clone())
            eng.optimizer_step(lr=1e-3)
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}       # right behind the step: the hook must wait for the update
        torch.cuda.synchronize()
        assert not pipelined or eng._side_stream() is None or eng._upd is None
        res.append((torch.stack(sums), eng.P32.clone(), eng.opt_m.clone(), eng.opt_v.clone(), eng.Pbf.clone(), sd))
    a, b = res
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    for k in a[5]:
        assert torch.equal(a[5][k], b[5][k]), k
    assert float((a[0][0] - a[0][2]).abs().max()) > 0                         # the steps did move the parameters


def test_packed_step_with_an_empty_sequence(ops, monkeypatch):
    """Edge of the row maps: a sequence with no visible encoder row at all (its decoder queries then see no cross-attention key:
    zero rows, like the padded kernels) and a decoder side that is just the SOS row; packed and padded steps must agree."""
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    e2w, w2e = load_vocab()
    B, S, d = 4, 256, 256
    cfg = BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                     encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0)
    m = PianoBartLM(PianoBart(cfg, e2w, w2e, precision='bf16'))
    randomize_params(m, 13)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(B, S, seed=4)]
    emask, dmask, loss_mask = emask.clone().float(), dmask.clone().float(), loss_mask.clone().float()
    emask[0] = 0                                                        # nothing of sequence 0 is visible to anybody
    dmask[0] = 0; dmask[0, 0] = 1
    loss_mask[0] = 0; loss_mask[0, 0] = 1
    emask[1, 100:] = 0; dmask[1, 50:] = 0; loss_mask[1, 50:] = 0
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    s0, g0, r0 = _step(eng, args, False, monkeypatch)
    s1, g1, r1 = _step(eng, args, True, monkeypatch)
    assert r1[0] < B * S and r1[1] < B * S, r1
    assert torch.isfinite(s1).all() and torch.isfinite(g1).all()
    assert torch.equal(s0[8:16], s1[8:16]) and torch.allclose(s0[:8], s1[:8], rtol=1e-3)
    for name, sl in eng.slots.items():
        a, b_ = g0[sl.off:sl.off + sl.numel], g1[sl.off:sl.off + sl.numel]
        assert float((a - b_).norm()) <= 2e-2 * float(a.norm()) + 1e-6, name


def test_packed_step_against_the_oracle_directly(ops, monkeypatch):
    """The step the bench times (bf16, dead rows dropped, last decoder layer on the loss rows) against the CPU oracle on the PADDED
    batch -- no dense HIP step in between. 2L / 256d / 4 heads, B = 6, S = 256, ragged PAD tails on both sides, sparse loss mask,
    dropout 0. Follows pretrain.py:112-118 (masked CE, sum / sum per head, e2w-order weights) and :159-196 (forward, backward).
    Bounds are ~2x the measured bf16-vs-f32 differences of this case (loss 2e-4, gradient norm 4e-3, named gradients 2-4e-2)."""
    from oracle import pianobart_oracle as O
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    e2w, w2e = load_vocab()
    B, S, d = 6, 256, 256
    kw = dict(max_position_embeddings=S, d_model=d, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
              encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0)
    m = PianoBartLM(PianoBart(BartConfig(**kw), e2w, w2e, precision='bf16')).train()
    randomize_params(m, 11)
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**kw), e2w, w2e)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(B, S, seed=9)
    rng = np.random.default_rng(3)
    Le, Ld = rng.integers(40, S + 1, size=B), rng.integers(40, S + 1, size=B)
    Le[0], Ld[1] = S, S
    emask, dmask, loss_mask = emask.clone().float(), dmask.clone().float(), loss_mask.clone().float()
    for b in range(B):
        emask[b, :Le[b]] = 1; emask[b, Le[b]:] = 0
        dmask[b, :Ld[b]] = 1; dmask[b, Ld[b]:] = 0
        loss_mask[b] = 0
        loss_mask[b, :Ld[b]] = torch.from_numpy((rng.random((Ld[b], 8)) < 0.15).astype(np.float32))
        loss_mask[b, 1, :] = 1
    # the oracle on the padded batch (CPU, seconds)
    yo = o(enc, dec, emask, dmask)
    total_o, *_ = O.pretrain_loss(yo, target, loss_mask, e2w)
    total_o.backward()
    go = {k: p.grad.double() for k, p in o.named_parameters() if p.grad is not None}
    gn_o = float(torch.sqrt(sum((g ** 2).sum() for g in go.values())))
    # the packed HIP step
    dev = lambda t: t.cuda()
    args = (ops.ids_to_i16(dev(enc)), ops.ids_to_i16(dev(dec)), ops.ids_to_i16(dev(target)), dev(loss_mask).contiguous(), dev(emask), dev(dmask))
    sums, G, rows = _step(eng, args, True, monkeypatch)
    Te, Td, T, Ts = rows
    assert Te < T and Td < T and Ts < Td, rows                          # it really packed, and the last layer ran on the loss rows only
    s = sums.double().cpu()
    w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
    loss = float(((s[0:8] / s[8:16]) * w).sum() / w.sum())
    assert torch.equal(s[8:16], loss_mask.double().sum((0, 1)))
    assert abs(loss - float(total_o)) / float(total_o) < 5e-4, (loss, float(total_o))
    views = {id(p): g for p, g in zip(eng.params, eng.grad_views_of(G))}
    named = {k: views[id(p)].double().cpu() for k, p in m.named_parameters() if id(p) in views}
    gn = float(torch.sqrt(sum((g ** 2).sum() for k, g in named.items() if k in go)))
    print('packed bf16 step vs oracle: loss %.6f / %.6f, grad norm %.5f / %.5f, rows %s' % (loss, float(total_o), gn, gn_o, rows))
    assert abs(gn - gn_o) / gn_o < 1e-2
    for k in ('pianobart.bart.encoder.layers.0.fc1.weight', 'pianobart.bart.decoder.layers.1.encoder_attn.q_proj.weight',
              'pianobart.bart.decoder.layers.1.fc2.weight', 'pianobart.bart.decoder.layers.0.self_attn.v_proj.weight', 'mask_lm.proj.3.weight',
              'pianobart.word_emb.3.lut.weight', 'pianobart.bart.decoder.layers.1.final_layer_norm.weight'):
        e = float((named[k] - go[k]).norm() / go[k].norm())
        print('   %-70s rel %.3e' % (k, e))
        assert e < 8e-2, (k, e)


def test_one_pass_backward_in_the_padded_step(ops, monkeypatch):
    """The padded step's head_dim-64 non-causal attention backward runs in one pass (csrc/pb_flash1.hip, engine._ATTN_BWD1 = 1): against the
    dQ + dK/dV kernel pair on the same batch the loss sums are identical (same forward) and every gradient slot agrees to bf16 noise:
    dK / dV are the same products in the same order, dQ is rounded once more (bf16 slab per 256-key block). Measured 5e-3 .. 9e-3
    relative on the slots at the end of the backward chain, where the rounding differences of every layer have accumulated."""
    from pianobart_amd import engine as E
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    e2w, w2e = load_vocab()
    B, S, d = 4, 512, 256
    cfg = BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                     encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.1)
    m = PianoBartLM(PianoBart(cfg, e2w, w2e, precision='bf16'))
    randomize_params(m, 13)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(B, S, seed=4)]
    emask = emask.clone().float(); emask[0] = 0; emask[1, 300:] = 0; emask[2, ::3] = 0          # no visible key | a PAD tail | scattered holes
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    monkeypatch.setattr(E, '_PACK_ROWS', 0)
    out = {}
    for mode in (0, 1):
        monkeypatch.setattr(E, '_ATTN_BWD1', mode)
        eng._seed = 5
        sums = eng.loss_and_grads(*args, train=True).clone()
        torch.cuda.synchronize()
        out[mode] = (sums, eng.G32.clone())
    assert torch.equal(out[0][0], out[1][0])
    assert torch.isfinite(out[1][1]).all()
    for name, sl in eng.slots.items():
        a, b_ = out[0][1][sl.off:sl.off + sl.numel], out[1][1][sl.off:sl.off + sl.numel]
        assert float((a - b_).norm()) <= 2e-2 * float(a.norm()) + 1e-6, name


def test_one_pass_backward_hands_sequences_beyond_6144_rows_to_the_kernel_pair(ops, monkeypatch):
    """the one-pass attention backward keeps a sequence's -lse / -delta tables in LDS and takes at most 6144 queries; the default
    dispatch (PB_ATTN_BWD1=2) must give longer sequences to the dQ + dK/dV pair instead of failing. S = 6400, dense and packed step."""
    from pianobart_amd import engine as E
    from pianobart_amd._lib import LIB, PBError
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    assert LIB.query('pb_flash_bwd1_supported', 6144, 6144, 64, 6144, 1) == 1 and LIB.query('pb_flash_bwd1_supported', 6400, 6400, 64, 6400, 1) == 0
    e2w, w2e = load_vocab()
    B, S, d = 1, 6400, 64
    cfg = BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=128, decoder_ffn_dim=128,
                     encoder_attention_heads=1, decoder_attention_heads=1, dropout=0.0)
    m = PianoBartLM(PianoBart(cfg, e2w, w2e, precision='bf16'))
    randomize_params(m, 3)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(B, S, seed=9, min_len=S - 40)]
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    out = {}
    for pack in (0, 1):
        for mode in (0, 2):                                               # never one-pass | the default
            monkeypatch.setattr(E, '_PACK_ROWS', pack)
            monkeypatch.setattr(E, '_ATTN_BWD1', mode)
            sums = eng.loss_and_grads(*args, train=True).clone()
            torch.cuda.synchronize()
            out[pack, mode] = (sums, eng.Gcur.clone())
        assert torch.equal(out[pack, 0][0], out[pack, 2][0]) and torch.equal(out[pack, 0][1], out[pack, 2][1])   # the same kernels ran
        assert torch.isfinite(out[pack, 2][1]).all() and float(out[pack, 2][1].abs().max()) > 0
    # the kernel itself still refuses the shape, loudly
    q = torch.zeros(B, S, 3 * d, device='cuda', dtype=torch.bfloat16)
    o = torch.zeros(B, S, d, device='cuda', dtype=torch.bfloat16)
    lse = torch.zeros(B, 1, S, device='cuda')
    with pytest.raises(PBError):
        ops.flash_bwd1((q, 0, 3 * d, S * 3 * d), (q, d, 3 * d, S * 3 * d), (q, 2 * d, 3 * d, S * 3 * d), (o, 0, d, S * d), o, lse, None,
                       (q, 0, 3 * d, S * 3 * d), (q, d, 3 * d, S * 3 * d), (q, 2 * d, 3 * d, S * 3 * d), torch.zeros(B, 1, S, device='cuda'),
                       B, 1, S, S, 64, 0.125, False)


@pytest.mark.parametrize('heads,d,tail_loss,hole,dropout,sparse', [(4, 256, False, False, 0.0, False), (4, 256, True, True, 0.1, True), (2, 256, True, False, 0.1, True),
                                                                   (4, 128, True, False, 0.1, True)])
def test_packed_step_of_the_split_bf16_instantiation(ops, monkeypatch, heads, d, tail_loss, hole, dropout, sparse):
    """Dead-row compaction under precision='bf16x3' (round 6: packed rows through pb_flash_*_x3_packed, head_dim 32 / 64 / 128): the packed step gives the
    padded step's loss sums and every gradient slot to 1e-4 (f32 storage: what differs is the summation order of the GEMM tiles and of the f32 atomics of
    the embedding gradients), dropout included, with loss rows outside the visible prefix and a hole in the encoder mask; and, dropout off, the CPU oracle's
    loss (1e-5) and every parameter gradient (2e-3 of the tensor's largest element) on the PADDED batch -- pretrain.py:112-118, 159-196."""
    from oracle import pianobart_oracle as O
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
    e2w, w2e = load_vocab()
    B, S = 6, 256
    kw = dict(max_position_embeddings=S, d_model=d, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
              encoder_attention_heads=heads, decoder_attention_heads=heads, dropout=dropout)
    m = PianoBartLM(PianoBart(BartConfig(**kw), e2w, w2e, precision='bf16x3'))
    randomize_params(m, 11)
    o = O.PianoBartLM(O.PianoBart(O.BartConfig(**kw), e2w, w2e)).train()
    o.load_state_dict(m.state_dict(), strict=True)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(B, S, seed=9)
    rng = np.random.default_rng(3)
    Le, Ld = rng.integers(40, S + 1, size=B), rng.integers(40, S + 1, size=B)
    Le[0], Ld[1] = S, S
    emask, dmask, loss_mask = emask.clone().float(), dmask.clone().float(), loss_mask.clone().float()
    for b in range(B):
        emask[b, :Le[b]] = 1; emask[b, Le[b]:] = 0
        if hole:
            emask[b, 3] = 0
        dmask[b, :Ld[b]] = 1; dmask[b, Ld[b]:] = 0
        loss_mask[b] = 0
        loss_mask[b, :Ld[b]] = torch.from_numpy((rng.random((Ld[b], 8)) < (0.15 if sparse else 1.1)).astype(np.float32))
        loss_mask[b, 1, :] = 1
        if tail_loss and Ld[b] + 5 < S:
            loss_mask[b, Ld[b] + 2, :] = 1
            loss_mask[b, S - 1, 2] = 1
    dev = lambda t: t.cuda()
    args = (ops.ids_to_i16(dev(enc)), ops.ids_to_i16(dev(dec)), ops.ids_to_i16(dev(target)), dev(loss_mask).contiguous(), dev(emask), dev(dmask))
    s0, g0, r0 = _step(eng, args, False, monkeypatch)
    s1, g1, r1 = _step(eng, args, True, monkeypatch)
    assert r0 == (B * S,) * 4 and r1[0] < B * S and r1[1] < B * S, (r0, r1)          # it really packed
    assert (r1[3] < r1[1]) if sparse else r1[3] == r1[1], r1
    assert torch.equal(s0[8:16], s1[8:16]) and torch.allclose(s0[:8], s1[:8], rtol=2e-5), (s0 - s1).abs().max()
    worst = ('', 0.0)
    for name, sl in eng.slots.items():
        a, b_ = g0[sl.off:sl.off + sl.numel], g1[sl.off:sl.off + sl.numel]
        worst = max(worst, (name, float((a - b_).norm()) / (float(a.norm()) + 1e-12)), key=lambda t: t[1])
    print('bf16x3 packed vs padded: worst gradient slot', worst, 'rows', r1)
    assert torch.isfinite(g1).all() and worst[1] < 1e-4, worst
    if dropout == 0.0:
        total_o, *_ = O.pretrain_loss(o(enc, dec, emask, dmask), target, loss_mask, e2w)
        total_o.backward()
        s = s1.double().cpu()
        w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
        loss = float(((s[0:8] / s[8:16]) * w).sum() / w.sum())
        assert abs(loss - float(total_o)) / float(total_o) < 1e-5, (loss, float(total_o))
        views = {id(p): g for p, g in zip(eng.params, eng.grad_views_of(g1))}
        scale = max(float(p.grad.abs().max()) for p in o.parameters() if p.grad is not None)
        for k, p in m.named_parameters():
            go = dict(o.named_parameters())[k].grad
            if go is None or id(p) not in views:
                continue
            err = float((views[id(p)].double().cpu() - go.double()).abs().max())
            assert err < 2e-3 * max(float(go.abs().max()), 1e-3 * scale), (k, err, float(go.abs().max()))
