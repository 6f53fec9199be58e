"""GPU: every HIP kernel of libpianobart_hip.so against a plain PyTorch fp32/fp64 reference of the
same op (called through the C ABI via pianobart_amd.ops)."""
import math

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


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('a_kc,b_kc', [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 384, 192), (200, 72, 104), (38, 768, 256), (1000, 1280, 96), (5, 8, 8)])
def test_gemm_layouts(ops, dt, a_kc, b_kc, M, N, K):
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.randn(N, K, device='cuda', generator=g) + 0.1 * torch.arange(N, device='cuda')[:, None] / N   # asymmetric
    ref = A.to(dt).double() @ B.to(dt).double().t()
    Am = (A if a_kc else A.t()).contiguous().to(dt)
    Bm = (B if b_kc else B.t()).contiguous().to(dt)
    C = torch.full((M, N), float('nan'), device='cuda', dtype=dt)
    ops.gemm(Am, Bm, C, M=M, N=N, K=K, dtype=ops.dtype_code(dt), a_kc=a_kc, b_kc=b_kc)
    torch.cuda.synchronize()
    assert _rel(C, ref) < (3e-6 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_identity_asymmetric(ops, dt):
    """A = I with asymmetric B catches a transposed C write (guide: cdna_hip_programming 3)."""
    n = 128
    A = torch.eye(n, device='cuda', dtype=dt)
    B = (torch.arange(n, device='cuda')[:, None] * 3 + torch.arange(n, device='cuda')[None, :] % 7).to(dt)  # B[n,k]
    C = torch.zeros(n, n, device='cuda', dtype=dt)
    ops.gemm(A, B, C, M=n, N=n, K=n, dtype=ops.dtype_code(dt))
    assert torch.equal(C.float(), B.float().t())


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_epilogues(ops, dt):
    g = torch.Generator(device='cuda').manual_seed(3)
    M, N, K = 192, 320, 128
    A = torch.randn(M, K, device='cuda', generator=g).to(dt)
    W = (torch.randn(N, K, device='cuda', generator=g) / math.sqrt(K)).to(dt)
    bias = torch.randn(N, device='cuda', generator=g)
    code = ops.dtype_code(dt)
    pre = (A.double() @ W.double().t() + bias.double()).requires_grad_(True)
    torch.nn.functional.gelu(pre).sum().backward()
    # bias + GELU (the activation's derivative is saved for the backward)
    C = torch.empty(M, N, device='cuda', dtype=dt); U = torch.empty_like(C)
    ops.gemm(A, W, C, M=M, N=N, K=K, dtype=code, bias=bias, gelu_aux_out=U)
    assert _rel(U, pre.grad) < TOL[dt] and _rel(C, torch.nn.functional.gelu(pre.detach())) < TOL[dt]
    # alpha + accumulate into f32 C
    C32 = torch.randn(M, N, device='cuda', generator=g)
    ref = C32.double() + 0.5 * (A.double() @ W.double().t())
    ops.gemm(A, W, C32, M=M, N=N, K=K, dtype=code, alpha=0.5, accum=True, c_f32=True)
    assert _rel(C32, ref) < TOL[dt]
    # multiply by the saved derivative
    u = torch.randn(M, N, device='cuda', generator=g).to(dt)
    C = torch.empty(M, N, device='cuda', dtype=dt)
    ops.gemm(A, W, C, M=M, N=N, K=K, dtype=code, gelu_grad_aux_in=u)
    assert _rel(C, (A.double() @ W.double().t()) * u.double()) < TOL[dt]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_batched_strided(ops, dt):
    """The unfused attention products: batch over (b,h) with head-sliced strides."""
    g = torch.Generator(device='cuda').manual_seed(5)
    B, H, S, hd = 2, 3, 72, 32
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, device='cuda', generator=g).to(dt)
    scores = torch.empty(B, H, S, S, device='cuda')
    code = ops.dtype_code(dt)
    ops.gemm(qkv, qkv, scores, M=S, N=S, K=hd, dtype=code, lda=3 * d, ldb=3 * d, ldc=S, c_f32=True, nb1=B, nb2=H,
             sA=(S * 3 * d, hd), sB=(S * 3 * d, hd), sC=(H * S * S, S * S), b_off=d)
    q = qkv[..., :d].reshape(B, S, H, hd).permute(0, 2, 1, 3).double()
    k = qkv[..., d:2 * d].reshape(B, S, H, hd).permute(0, 2, 1, 3).double()
    v = qkv[..., 2 * d:].reshape(B, S, H, hd).permute(0, 2, 1, 3).double()
    assert _rel(scores, q @ k.transpose(2, 3)) < TOL[dt]
    P = torch.softmax(scores, -1).to(dt)
    ctx = torch.empty(B, S, d, device='cuda', dtype=dt)
    ops.gemm(P, qkv, ctx, M=S, N=hd, K=S, dtype=code, b_kc=False, lda=S, ldb=3 * d, ldc=d, nb1=B, nb2=H,
             sA=(H * S * S, S * S), sB=(S * 3 * d, hd), sC=(S * d, hd), b_off=2 * d)
    ref = (P.double() @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    assert _rel(ctx, ref) < TOL[dt]


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('d', [128, 768, 1024, 64])
@pytest.mark.parametrize('p', [0.0, 0.1])
def test_add_ln_fwd_bwd(ops, dt, d, p):
    g = torch.Generator(device='cuda').manual_seed(d)
    T = 301
    res = torch.randn(T, d, device='cuda', generator=g).to(dt)
    a = torch.randn(T, d, device='cuda', generator=g).to(dt)
    w = 1 + 0.2 * torch.randn(d, device='cuda', generator=g)
    b = 0.2 * torch.randn(d, device='cuda', generator=g)
    dy = torch.randn(T, d, device='cuda', generator=g).to(dt)
    y = torch.empty(T, d, device='cuda', dtype=dt)
    mean = torch.empty(T, device='cuda'); rstd = torch.empty(T, device='cuda')
    seed, site = 1234567, 5
    ops.add_ln_fwd(res, a, w, b, y, mean, rstd, 1e-5, seed, site, p)
    # recover the dropout mask by running the same site on (0, ones) with identity LN statistics
    if p > 0:
        ones = torch.ones(T, d, device='cuda', dtype=dt); zeros = torch.zeros_like(ones)
        y1 = torch.empty_like(ones); m1 = torch.empty(T, device='cuda'); r1 = torch.empty(T, device='cuda')
        ops.add_ln_fwd(zeros, ones, torch.ones(d, device='cuda'), torch.zeros(d, device='cuda'), y1, m1, r1, 1e-5, seed, site, p)
        # z = mask/(1-p); y1 = (z-mean)*rstd  -> z = y1/rstd + mean
        z = y1.float() / r1[:, None] + m1[:, None]
        mask = (z > 0.5).double() / (1 - p)
        frac = float((mask == 0).double().mean())
        assert abs(frac - p) < 0.01
    else:
        mask = torch.ones(T, d, device='cuda', dtype=torch.double)
    rd = res.double().requires_grad_(True); ad = a.double().requires_grad_(True)
    wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(rd + ad * mask, (d,), wd, bd, 1e-5)
    yr.backward(dy.double())
    assert _rel(y, yr) < TOL[dt]
    dres = torch.empty(T, d, device='cuda', dtype=dt); da = torch.empty_like(dres)
    dg = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda'); dba = torch.zeros(d, device='cuda')
    partials = torch.empty(int(ops.LIB.query('pb_ln_partials_floats', d)), device='cuda')
    ops.add_ln_bwd(dy, res, a, w, mean, rstd, dres, da, dg, db, dba, partials, False, seed, site, p)
    assert _rel(dres, rd.grad) < TOL[dt] * 2
    assert _rel(da, ad.grad) < TOL[dt] * 2
    assert _rel(dg, wd.grad) < TOL[dt] * 2 and _rel(db, bd.grad) < TOL[dt] * 2
    assert _rel(dba, ad.grad.sum(0)) < TOL[dt] * 4


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('d', [128, 768])
def test_embed_ln_fwd_bwd(ops, dt, d):
    g = torch.Generator(device='cuda').manual_seed(9)
    B, S = 3, 40
    T = B * S
    ids = torch.stack([torch.randint(0, n, (T,), device='cuda', generator=g) for n in ops.SEG_SIZES], dim=1)
    P = torch.randn(ops.VOCAB, d, device='cuda', generator=g)
    lb = torch.randn(d, device='cuda', generator=g); pos = torch.randn(S + 2, d, device='cuda', generator=g)
    w = 1 + 0.2 * torch.randn(d, device='cuda', generator=g); b = 0.2 * torch.randn(d, device='cuda', generator=g)
    ids16 = ops.ids_to_i16(ids)
    assert torch.equal(ids16.long(), ids)
    y = torch.empty(T, d, device='cuda', dtype=dt); mean = torch.empty(T, device='cuda'); rstd = torch.empty(T, device='cuda')
    ops.embed_ln_fwd(ids16, P, lb, pos, w, b, y, mean, rstd, S, 1e-5, 0, 0, 0.0)
    Pd = P.double().requires_grad_(True); lbd = lb.double().requires_grad_(True); posd = pos.double().requires_grad_(True)
    wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    off = torch.tensor(ops.SEG_OFF[:8], device='cuda')
    z = Pd[(ids + off).reshape(-1)].reshape(T, 8, d).sum(1) + lbd + posd[2:2 + S].repeat(B, 1)
    yr = torch.nn.functional.layer_norm(z, (d,), wd, bd, 1e-5)
    assert _rel(y, yr) < TOL[dt]
    dy = torch.randn(T, d, device='cuda', generator=g).to(dt)
    dy[7] = 0                                                    # a row without gradient (skip path)
    yr.backward(dy.double())
    dP = torch.zeros_like(P); dpos = torch.zeros_like(pos); dlb = torch.zeros(d, device='cuda')
    dg = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda')
    partials = torch.empty(int(ops.LIB.query('pb_ln_partials_floats', d)), device='cuda')
    ops.embed_ln_bwd(dy, ids16, P, lb, pos, w, mean, rstd, dP, dpos, dlb, dg, db, partials, S, 0, 0, 0.0)
    assert _rel(dP, Pd.grad) < 1e-4 and _rel(dpos, posd.grad) < 1e-4 and _rel(dlb, lbd.grad) < 1e-4
    assert _rel(dg, wd.grad) < 1e-4 and _rel(db, bd.grad) < 1e-4


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('causal', [False, True])
def test_softmax_fwd_bwd(ops, dt, causal):
    g = torch.Generator(device='cuda').manual_seed(2)
    B, H, Sq, Sk = 2, 3, 50, 50
    scores = torch.randn(B, H, Sq, Sk, device='cuda', generator=g) * 3
    km = (torch.rand(B, Sk, device='cuda', generator=g) > 0.3).float()
    km[1, :] = 0                                                  # batch 1: nothing visible -> zero rows
    km[0, 0] = 0                                                  # causal row 0 of batch 0 sees nothing
    P = torch.empty(B, H, Sq, Sk, device='cuda', dtype=dt)
    ops.softmax_fwd(scores, km, P, B, H, Sq, Sk, 0.25, causal)
    vis = (km != 0)[:, None, None, :].expand(B, H, Sq, Sk)
    if causal:
        vis = vis & torch.ones(Sq, Sk, dtype=torch.bool, device='cuda').tril()
    s = (scores.double() * 0.25).masked_fill(~vis, float('-inf'))
    ref = torch.where(vis.any(-1, keepdim=True), torch.softmax(s, -1), torch.zeros_like(s))
    ref = torch.nan_to_num(ref, nan=0.0)
    assert float((P.double() - ref).abs().max()) < (1e-6 if dt == torch.float32 else 4e-3)
    dP = torch.randn(B, H, Sq, Sk, device='cuda', generator=g)
    dS = torch.empty_like(P)
    ops.softmax_bwd(dP, P, dS, B * H * Sq, Sk, 0.25)
    Pd = P.double()
    refd = 0.25 * Pd * (dP.double() - (dP.double() * Pd).sum(-1, keepdim=True))
    assert float((dS.double() - refd).abs().max()) < (1e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_ce_fwd_bwd(ops, dt):
    g = torch.Generator(device='cuda').manual_seed(4)
    T, V = 333, ops.VOCAB
    logits = torch.randn(T, V, device='cuda', generator=g) * 2
    logits[5, 0] = logits[5, 1] = 50.0                           # tie: argmax must pick the lower index
    target = torch.stack([torch.randint(0, n, (T,), device='cuda', generator=g) for n in ops.SEG_SIZES], dim=1)
    m = (torch.rand(T, 8, device='cuda', generator=g) < 0.3).float()
    m[:, 5] = 0; m[0, 5] = 1
    w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], device='cuda', dtype=torch.float32)
    counts = torch.empty(8, device='cuda'); coef = torch.empty(8, device='cuda')
    partials = torch.empty(int(ops.LIB.query('pb_ce_partials_floats')), device='cuda')
    ops.mask_count(m, counts, partials); ops.loss_coef(counts, w, coef)
    assert torch.allclose(counts, m.sum(0))
    sums = torch.zeros(24, device='cuda')
    partials = torch.empty(int(ops.LIB.query('pb_ce_partials_floats')), device='cuda')
    dl = torch.empty(T, V, device='cuda', dtype=dt); am = torch.empty(T, 8, device='cuda', dtype=torch.int16)
    ops.ce_fwd_bwd(logits, target.to(torch.int16), m, sums, partials, coef, dl, am)
    ld = logits.double().requires_grad_(True)
    total = 0
    for i in range(8):
        seg = ld[:, ops.SEG_OFF[i]:ops.SEG_OFF[i + 1]]
        ce = torch.nn.functional.cross_entropy(seg, target[:, i], reduction='none')
        li = (ce * m[:, i].double()).sum() / m[:, i].double().sum()
        assert abs(float(sums[i]) / float(sums[8 + i]) - float(li)) < 1e-5 * max(1, abs(float(li)))
        ok = ((seg.argmax(-1) == target[:, i]).double() * m[:, i].double()).sum()
        assert abs(float(sums[16 + i]) - float(ok)) < 1e-3
        assert torch.equal(am[:, i].long(), torch.from_numpy(np.argmax(seg.detach().cpu().numpy(), -1)).cuda())
        total = total + li * w[i].double()
    total = total / w.double().sum()
    total.backward()
    assert _rel(dl, ld.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    assert int(am[5, 0]) == 0


def test_optimizer_kernels(ops):
    from oracle import pianobart_oracle as O
    g = torch.Generator(device='cuda').manual_seed(8)
    n = 100003
    n_pad = (n + 3) // 4 * 4
    p = torch.randn(n_pad, device='cuda', generator=g)[:n]; gr = torch.randn(n_pad, device='cuda', generator=g)[:n] * 0.1
    m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda'); sh = torch.empty(n, device='cuda', dtype=torch.bfloat16)
    partials = torch.empty(int(ops.LIB.query('pb_norm_partials_floats')), device='cuda')
    sq = torch.empty(1, device='cuda'); coef = torch.empty(1, device='cuda')
    ops.grad_sqnorm(gr, partials, sq)
    assert abs(float(sq) - float((gr.double() ** 2).sum())) / float(sq) < 1e-6
    ops.clip_coef(sq, 3.0, 1.0, coef)
    gc = gr.cpu().clone(); tot = O.clip_grad_norm([gc], 3.0)
    pc, mc, vc = p.cpu().clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        ops.adamw_step(p, gr, m, v, sh, coef, 2e-5, 0.9, 0.999, 1e-6, 0.01, step)
        O.hf_adamw_step([pc], [gc], [mc], [vc], step=step, lr=2e-5)
    assert _rel(p.cpu(), pc) < 1e-6 and _rel(m.cpu(), mc) < 1e-5 and _rel(v.cpu(), vc) < 1e-4
    assert torch.equal(sh.cpu(), p.cpu().to(torch.bfloat16))
    x = torch.randn(1027, device='cuda', generator=g); xb = torch.empty(1027, device='cuda', dtype=torch.bfloat16); xf = torch.empty(1027, device='cuda')
    ops.cast_f32_to_bf16(x, xb); ops.cast_bf16_to_f32(xb, xf)
    assert torch.equal(xb, x.to(torch.bfloat16)) and torch.equal(xf, xb.float())
    # degenerate norms follow torch.nn.utils.clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1) hands a NaN on, an infinite norm gives 0
    for bad, want in ((float('nan'), float('nan')), (float('inf'), 0.0), (0.0, 1.0)):
        sq.fill_(bad); ops.clip_coef(sq, 3.0, 1.0, coef)
        ref = torch.clamp(torch.tensor(3.0) / (torch.tensor(bad).sqrt() + 1e-6), max=1.0)
        assert (torch.isnan(coef).item() and torch.isnan(ref).item()) if want != want else float(coef) == float(ref) == want


@pytest.mark.parametrize('hd', [32, 64, 96, 128])
@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('S', [64, 200, 136, 384])
@pytest.mark.parametrize('generic', [False, True])
def test_flash_attention_fwd_bwd(ops, hd, causal, S, generic):
    """Fused attention (bf16) vs an fp64 reference incl. key-padding masks, causal, ragged S and zero rows."""
    if hd == 96 and generic:
        pytest.skip('head_dim 96 exists in the pipelined kernel family only')
    g = torch.Generator(device='cuda').manual_seed(hd + S)
    B, H = 2, 3
    d = H * hd
    qkv = (torch.randn(B, S, 3 * d, device='cuda', generator=g) * 1.5).to(torch.bfloat16)
    km = (torch.rand(B, S, device='cuda', generator=g) > 0.25).float()
    km[0, 0] = 0                       # causal row 0 of batch 0 sees nothing -> zero row
    km[1, S // 2:] = 0                 # padded tail
    out = torch.full((B, S, d), float('nan'), device='cuda', dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device='cuda')
    scale = hd ** -0.5
    sl = lambda off: (qkv, off, 3 * d, S * 3 * d)
    kmax = None
    if hd in (64, 96, 128) and not generic:      # tile skipping past the last visible key (pipelined kernel family)
        kmax = torch.empty(B, dtype=torch.int32, device='cuda')
        ops.key_extent(km, kmax)
        assert kmax.tolist() == [int(km[b].nonzero().max()) + 1 for b in range(B)]
    ops.flash_fwd(sl(0), sl(d), sl(2 * d), (out, 0, d, S * d), lse, km, B, H, S, S, hd, scale, causal, force_generic=generic, kmax=kmax)
    qd = qkv.double().requires_grad_(True)
    q = qd[..., :d].reshape(B, S, H, hd).permute(0, 2, 1, 3)
    k = qd[..., d:2 * d].reshape(B, S, H, hd).permute(0, 2, 1, 3)
    v = qd[..., 2 * d:].reshape(B, S, H, hd).permute(0, 2, 1, 3)
    vis = (km != 0)[:, None, None, :].expand(B, H, S, S)
    if causal:
        vis = vis & torch.ones(S, S, dtype=torch.bool, device='cuda').tril()
    s = (q @ k.transpose(2, 3) * scale).masked_fill(~vis, float('-inf'))
    p = torch.where(vis.any(-1, keepdim=True), torch.softmax(s, -1), torch.zeros_like(s))
    p = torch.nan_to_num(p, nan=0.0)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    # the forward prescales Q by scale * log2(e) in bf16 (one more rounding of the scores, as the backward kernels always did)
    assert float((out.double() - ref).abs().max()) < 4e-2
    has = vis.any(-1)
    ref_lse = torch.logsumexp(s, -1)
    assert float((lse.double() - ref_lse)[has].abs().max()) < 2e-2
    dout = torch.randn(B, S, d, device='cuda', generator=g).to(torch.bfloat16)
    ref.backward(dout.double())
    dqkv = torch.full((B, S, 3 * d), float('nan'), device='cuda', dtype=torch.bfloat16)
    delta = torch.empty(B, H, S, device='cuda')
    dsl = lambda off: (dqkv, off, 3 * d, S * 3 * d)
    fused_bias = hd in (64, 96, 128) and not generic           # bias gradients of the q/k/v projections from the epilogue registers
    db, dbws = None, None
    if fused_bias:
        from pianobart_amd._lib import LIB
        db = [torch.full((d,), 0.25, device='cuda') for _ in range(3)]
        dbws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, S, S, hd)), device='cuda')
    ops.flash_bwd(sl(0), sl(d), sl(2 * d), (out, 0, d, S * d), dout, lse, km, dsl(0), dsl(d), dsl(2 * d), delta, B, H, S, S, hd, scale, causal, force_generic=generic, kmax=kmax,
                  dbias=db, dbias_ws=dbws)
    gref = qd.grad
    err = float((dqkv.double() - gref).abs().max() / gref.abs().max())
    assert err < 3e-2, err
    if fused_bias:
        cs = gref.reshape(B * S, 3 * d).sum(0)
        for i in range(3):
            assert float((db[i].double() - 0.25 - cs[i * d:(i + 1) * d]).abs().max() / cs.abs().max()) < 2e-2, i


@pytest.mark.parametrize('a_kc,b_kc', [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize('M,N,K,splitk', [(128, 128, 64, 1), (256, 384, 192, 1), (200, 136, 128, 1), (1000, 1280, 256, 1),
                                          (768, 384, 2048, 4), (264, 72, 1024, 8), (3072, 768, 4096, 2), (1000, 520, 448, 3),
                                          (512, 256, 128, 1)])
@pytest.mark.parametrize('tile256', [False, True])
@pytest.mark.parametrize('Mx', [1, 5])
def test_gemm2_fast_path_matches_generic(ops, a_kc, b_kc, M, N, K, splitk, tile256, Mx):
    M = M * Mx            # Mx = 5 reaches the 256x128 tile (M >= 1024) on the larger shapes
    """bf16 direct-to-LDS / transposed-read kernel (+ split-K slabs) vs fp64 and vs the generic kernel."""
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.randn(N, K, device='cuda', generator=g) + 0.1 * torch.arange(N, device='cuda')[:, None] / N
    dt = torch.bfloat16
    ref = A.to(dt).double() @ B.to(dt).double().t()
    Am = (A if a_kc else A.t()).contiguous().to(dt)
    Bm = (B if b_kc else B.t()).contiguous().to(dt)
    C2 = torch.full((M, N), float('nan'), device='cuda')
    C1 = torch.full((M, N), float('nan'), device='cuda')
    slabs = torch.empty(splitk * M * N, device='cuda') if splitk > 1 else None
    ops.gemm(Am, Bm, C2, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=True, splitk=splitk, slabs=slabs, tile256=tile256)
    ops.gemm(Am, Bm, C1, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, c_f32=True, force_v1=True)
    assert _rel(C2, ref) < 1e-5 and _rel(C1, ref) < 1e-5
    # bf16 output + epilogues on the fast path
    bias = torch.randn(N, device='cuda', generator=g)
    Cb = torch.empty(M, N, device='cuda', dtype=dt); U = torch.empty_like(Cb)
    ops.gemm(Am, Bm, Cb, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, bias=bias, alpha=0.125, gelu_aux_out=U, tile256=tile256)
    pre = (0.125 * ref + bias.double()).requires_grad_(True)
    torch.nn.functional.gelu(pre).sum().backward()
    assert _rel(U, pre.grad) < 1e-2 and _rel(Cb, torch.nn.functional.gelu(pre.detach())) < 1e-2
    acc0 = torch.randn(M, N, device='cuda', generator=g).to(dt)
    acc = acc0.clone()
    ops.gemm(Am, Bm, acc, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=a_kc, b_kc=b_kc, alpha=0.01, accum=True, tile256=tile256)
    assert _rel(acc, acc0.double() + 0.01 * ref) < 1e-2


def test_onehot_route_matches_atomic_scatter(ops):
    """dP via Onehot^T dz (MFMA, split-K) and dpos via batch_sum == the atomic scatter-add path."""
    g = torch.Generator(device='cuda').manual_seed(12)
    B, S, d = 4, 32, 128
    T = B * S
    ids = torch.stack([torch.randint(0, n, (T,), device='cuda', generator=g) for n in ops.SEG_SIZES], dim=1)
    ids16 = ops.ids_to_i16(ids)
    oh = torch.empty(T, ops.VOCAB, device='cuda', dtype=torch.bfloat16)
    ops.onehot_build(ids16, oh)
    ref = torch.zeros(T, ops.VOCAB, device='cuda')
    ref.scatter_(1, ids + torch.tensor(ops.SEG_OFF[:8], device='cuda'), 1.0)
    assert torch.equal(oh.float(), ref)
    dz = torch.randn(T, d, device='cuda', generator=g).to(torch.bfloat16)
    dP = torch.empty(ops.VOCAB, d, device='cuda')
    slabs = torch.empty(2 * ops.VOCAB * d, device='cuda')
    ops.gemm(oh, dz, dP, M=ops.VOCAB, N=d, K=T, dtype=ops.PB_BF16, a_kc=False, b_kc=False, lda=ops.VOCAB, ldb=d, ldc=d, c_f32=True,
             splitk=2, slabs=slabs, tile256=True)
    assert _rel(dP, ref.double().t() @ dz.double()) < 1e-5
    out = torch.ones(S, d, device='cuda')
    ops.batch_sum(dz, out, B, S * d)
    assert _rel(out, 1 + dz.double().reshape(B, S, d).sum(0)) < 1e-5


@pytest.mark.parametrize('M,N,K,dt', [(4096, 768, 256, torch.bfloat16), (520, 1280, 128, torch.bfloat16), (192, 320, 128, torch.bfloat16), (192, 320, 128, torch.float32)])
def test_gemm_fused_column_sums(ops, M, N, K, dt):
    """colsum_out += column sums of the stored C (bias gradient), from the 256x256 kernel's epilogue registers (first two shapes:
    interior and ragged tiles) or by the streaming pass behind the other kernels."""
    from pianobart_amd._lib import LIB
    g = torch.Generator(device='cuda').manual_seed(M + N)
    A = torch.randn(M, K, device='cuda', generator=g).to(dt)
    W = (torch.randn(K, N, device='cuda', generator=g) / math.sqrt(K)).to(dt)       # NN layout, like dgrad
    aux = torch.randn(M, N, device='cuda', generator=g).to(dt)
    C = torch.empty(M, N, device='cuda', dtype=dt)
    cs = torch.full((N,), 0.5, device='cuda')
    ws = torch.empty(int(LIB.query('pb_gemm_colsum_ws_floats', M, N)), device='cuda')
    ops.gemm(A, W, C, M=M, N=N, K=K, dtype=ops.dtype_code(dt), b_kc=False, ldb=N, gelu_grad_aux_in=aux, colsum_out=cs, colsum_ws=ws)
    ref = (A.double() @ W.double()) * aux.double()
    assert _rel(C, ref) < TOL[dt]
    assert _rel(cs, 0.5 + ref.sum(0)) < (2e-3 if dt == torch.bfloat16 else 1e-5)


def test_deferred_reductions_match_immediate(ops):
    """K15: between pb_defer_begin and pb_defer_flush the LayerNorm-backward and GEMM column-sum producers keep their partial rows in
    the arena and ONE launch reduces them; results must be bit-identical to the immediate reductions, also when the descriptor table
    or the arena is too small (those calls fall back to the immediate path) and for outputs that are not 16-byte aligned."""
    from pianobart_amd._lib import LIB
    dt, d, T = torch.bfloat16, 768, 2048
    g = torch.Generator(device='cuda').manual_seed(77)
    mk = lambda *s: torch.randn(*s, device='cuda', generator=g)
    cases = []
    for k in range(3):
        res, a, dy = mk(T, d).to(dt), mk(T, d).to(dt), mk(T, d).to(dt)
        w = 1 + 0.2 * mk(d)
        y = torch.empty(T, d, device='cuda', dtype=dt); mean = torch.empty(T, device='cuda'); rstd = torch.empty(T, device='cuda')
        ops.add_ln_fwd(res, a, w, torch.zeros(d, device='cuda'), y, mean, rstd, 1e-5, 99, k, 0.1)
        cases.append((dy, res, a, w, mean, rstd, k))
    A, W = mk(T, 256).to(dt), (mk(256, 512) / 16).to(dt)
    aux = mk(T, 512).to(dt)
    part = torch.empty(int(LIB.query('pb_ln_partials_floats', d)), device='cuda')
    csw = torch.empty(int(LIB.query('pb_gemm_colsum_ws_floats', T, 512)), device='cuda')

    def run(arena_floats, table_entries, misalign=False):
        outs = []
        # one flat output buffer; `misalign` shifts every output vector by one float (no 16-byte alignment -> immediate path)
        flat = torch.zeros(16 * 1024 + 1, device='cuda')
        pos = [1 if misalign else 0]

        def vec(n):
            v = flat[pos[0]:pos[0] + n]
            pos[0] += ((n + 3) // 4) * 4
            outs.append(v)
            return v
        arena = torch.empty(max(arena_floats, 4), device='cuda')
        table = torch.empty(max(table_entries, 1) * int(LIB.query('pb_defer_desc_bytes')), dtype=torch.uint8, device='cuda')
        if arena_floats:
            ops.defer_begin(arena, table)
        dres = torch.empty(T, d, device='cuda', dtype=dt); da = torch.empty_like(dres)
        for dy, res, a, w, mean, rstd, k in cases:
            ops.add_ln_bwd(dy, res, a, w, mean, rstd, dres, da, vec(d), vec(d), vec(d), part, False, 99, k, 0.1)
        C = torch.empty(T, 512, device='cuda', dtype=dt)
        ops.gemm(A, W, C, M=T, N=512, K=256, dtype=ops.PB_BF16, b_kc=False, ldb=512, gelu_grad_aux_in=aux, colsum_out=vec(512), colsum_ws=csw)
        if arena_floats:
            ops.defer_flush()
        torch.cuda.synchronize()
        return [o.clone() for o in outs]

    ref = run(0, 0)
    big = 3 * part.numel() + csw.numel() + 64
    for arena_floats, entries, mis in [(big, 16, False), (big, 2, False), (part.numel() + 8, 16, False), (big, 16, True), (big, 16, False)]:
        got = run(arena_floats, entries, mis)
        for r, o in zip(ref, got):
            assert torch.equal(r, o), (arena_floats, entries, mis, float((r - o).abs().max()))
    assert all(float(r.abs().max()) > 0 for r in ref)


@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('S', [64, 200, 333, 520, 1100])
@pytest.mark.parametrize('layout', ['qkv', 'cross'])
def test_one_pass_attention_backward(ops, causal, S, layout):
    """pb_flash_bwd1 (head_dim 64: 256 keys per workgroup, dK / dV in accumulators, dQ through bf16 slabs) vs an fp64 reference:
    key-padding masks with holes, a padded tail, a sample without any visible key, causal, sequence lengths that are no multiple of any
    tile (1, 2, 3 and 5 key blocks), q / k / v inside one qkv buffer or in separate q and (wide) kv buffers like the cross-attention of
    the stacked K / V projection; the q / k / v bias gradients come from the same launch. Bounds as for the kernel pair."""
    from pianobart_amd._lib import LIB
    hd, B, H = 64, 3, 2
    if causal and layout == 'cross':
        pytest.skip('the decoder self-attention is the only causal call')
    g = torch.Generator(device='cuda').manual_seed(S + 7 * causal)
    d = H * hd
    scale = hd ** -0.5
    if layout == 'qkv':
        buf = (torch.randn(B, S, 3 * d, device='cuda', generator=g) * 1.5).to(torch.bfloat16)
        ql, kl, vl = (buf, 0, 3 * d, S * 3 * d), (buf, d, 3 * d, S * 3 * d), (buf, 2 * d, 3 * d, S * 3 * d)
        leaf = buf
    else:
        qb = (torch.randn(B, S, d, device='cuda', generator=g) * 1.5).to(torch.bfloat16)
        kvb = (torch.randn(B, S, 6 * d, device='cuda', generator=g) * 1.5).to(torch.bfloat16)        # layer 1 of three stacked K | V projections
        ql, kl, vl = (qb, 0, d, S * d), (kvb, 2 * d, 6 * d, S * 6 * d), (kvb, 3 * d, 6 * d, S * 6 * d)
    km = (torch.rand(B, S, device='cuda', generator=g) > 0.25).float()
    km[0, 0] = 0
    km[1, S // 2:] = 0
    km[2] = 0                                                             # nothing visible: zero rows, zero gradients
    kmax = torch.empty(B, dtype=torch.int32, device='cuda')
    ops.key_extent(km, kmax)
    out = torch.empty(B, S, d, device='cuda', dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device='cuda')
    ops.flash_fwd(ql, kl, vl, (out, 0, d, S * d), lse, km, B, H, S, S, hd, scale, causal, kmax=kmax)
    if layout == 'qkv':
        qd = buf.double().requires_grad_(True)
        q4, k4, v4 = (qd[..., i * d:(i + 1) * d].reshape(B, S, H, hd).permute(0, 2, 1, 3) for i in range(3))
    else:
        qd, kvd = qb.double().requires_grad_(True), kvb.double().requires_grad_(True)
        q4 = qd.reshape(B, S, H, hd).permute(0, 2, 1, 3)
        k4, v4 = (kvd[..., i * d:(i + 1) * d].reshape(B, S, H, hd).permute(0, 2, 1, 3) for i in (2, 3))
    vis = (km != 0)[:, None, None, :].expand(B, H, S, S)
    if causal:
        vis = vis & torch.ones(S, S, dtype=torch.bool, device='cuda').tril()
    s = (q4 @ k4.transpose(2, 3) * scale).masked_fill(~vis, float('-inf'))
    p = torch.nan_to_num(torch.where(vis.any(-1, keepdim=True), torch.softmax(s, -1), torch.zeros_like(s)), nan=0.0)
    ref = (p @ v4).permute(0, 2, 1, 3).reshape(B, S, d)
    dout = torch.randn(B, S, d, device='cuda', generator=g).to(torch.bfloat16)
    ref.backward(dout.double())
    delta = torch.empty(B, H, S, device='cuda')
    db = [torch.full((d,), 0.25, device='cuda') for _ in range(3)]
    dbws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, S, S, hd)), device='cuda')
    if layout == 'qkv':
        dbuf = torch.full((B, S, 3 * d), float('nan'), device='cuda', dtype=torch.bfloat16)
        dq, dk, dv = (dbuf, 0, 3 * d, S * 3 * d), (dbuf, d, 3 * d, S * 3 * d), (dbuf, 2 * d, 3 * d, S * 3 * d)
    else:
        dqb = torch.full((B, S, d), float('nan'), device='cuda', dtype=torch.bfloat16)
        dkvb = torch.zeros(B, S, 6 * d, device='cuda', dtype=torch.bfloat16)
        dq, dk, dv = (dqb, 0, d, S * d), (dkvb, 2 * d, 6 * d, S * 6 * d), (dkvb, 3 * d, 6 * d, S * 6 * d)
    ops.flash_bwd1(ql, kl, vl, (out, 0, d, S * d), dout, lse, km, dq, dk, dv, delta, B, H, S, S, hd, scale, causal, kmax=kmax, dbias=db, dbias_ws=dbws)
    if layout == 'qkv':
        got, want = dbuf.double(), qd.grad
    else:
        got = torch.cat([dqb.double(), dkvb[..., 2 * d:4 * d].double()], -1)
        want = torch.cat([qd.grad, kvd.grad[..., 2 * d:4 * d]], -1)
        assert float(dkvb[..., :2 * d].abs().max()) == 0.0 and float(dkvb[..., 4 * d:].abs().max()) == 0.0       # the neighbours' columns are untouched
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 3e-2, err
    assert float(got[2].abs().max()) == 0.0                               # the sample without a visible key
    cs = want.reshape(B * S, 3 * d).sum(0)
    for i in range(3):
        assert float((db[i].double() - 0.25 - cs[i * d:(i + 1) * d]).abs().max() / cs.abs().max()) < 2e-2, i
    dsum = (dout.double() * out.double()).reshape(B, S, H, hd).sum(-1).permute(0, 2, 1)
    assert float((delta.double() - dsum).abs().max()) < 1e-4 * max(1.0, float(dsum.abs().max()))


@pytest.mark.parametrize('causal', [False, True])
def test_dispatch_order_changes_no_result(ops, causal):
    """bh_order (the order in which a packed attention grid takes the (batch, head) pairs, rowpack.dispatch_order: longest first, dealt in
    snake order over the XCDs) only re-orders the work: forward, kernel-pair backward and one-pass backward give bit-identical outputs
    with and without it. Ragged packed batch, head_dim 64; non-causal (encoder / cross form) and causal (the decoder's form: ordered too since
    round 6, rowpack.ORDER_CAUSAL)."""
    from pianobart_amd._lib import LIB
    from pianobart_amd.rowpack import dispatch_order
    hd, B, H, S = 64, 8, 4, 600
    d = H * hd
    gen = torch.Generator().manual_seed(3)
    qlen = torch.randint(S // 3, S + 1, (B,), generator=gen)
    kvis = (qlen - torch.randint(0, 30, (B,), generator=gen)).clamp(min=1)
    off = torch.cat([torch.zeros(1, dtype=torch.long), qlen.cumsum(0)[:-1]])
    T = int(qlen.sum())
    i32 = lambda x: x.to(torch.int32).cuda()
    order = torch.from_numpy(dispatch_order((qlen * kvis).numpy(), H)).cuda()
    assert sorted(order.tolist()) == list(range(B * H))
    g = torch.Generator(device='cuda').manual_seed(1)
    qkv = (torch.randn(T, 3 * d, device='cuda', generator=g) * 0.7).to(torch.bfloat16)
    dout = torch.randn(T, d, device='cuda', generator=g).to(torch.bfloat16)
    q, k, v = (qkv, 0, 3 * d), (qkv, d, 3 * d), (qkv, 2 * d, 3 * d)
    res = []
    for o in (None, order):
        rows = ops.PackedRows(i32(off), i32(qlen), i32(off), i32(qlen), i32(kvis), int(qlen.max()), int(qlen.max()), 'dec' if causal else 'enc', order=o)
        out = torch.full((T, d), float('nan'), device='cuda', dtype=torch.bfloat16)
        lse = torch.zeros(B, H, rows.Sq_max, device='cuda')
        ops.flash_fwd_packed(q, k, v, (out, 0, d), lse, rows, B, H, hd, hd ** -0.5, causal)
        grads = []
        for one in (False, True):
            dqkv = torch.full((T, 3 * d), float('nan'), device='cuda', dtype=torch.bfloat16)
            delta = torch.zeros(B, H, rows.Sq_max, device='cuda')
            db = [torch.zeros(d, device='cuda') for _ in range(3)]
            ws = torch.empty(int(LIB.query('pb_flash_bias_ws_floats', B, H, rows.Sq_max, rows.Sk_max, hd)), device='cuda')
            args = (q, k, v, (out, 0, d), dout, lse, (dqkv, 0, 3 * d), (dqkv, d, 3 * d), (dqkv, 2 * d, 3 * d), delta, rows, B, H, hd, hd ** -0.5, causal)
            if one:
                ops.flash_bwd1_packed(*args, T, dbias=db, dbias_ws=ws)
            else:
                ops.flash_bwd_packed(*args, dbias=db, dbias_ws=ws)
            torch.cuda.synchronize()
            grads += [dqkv, torch.stack(db)]
        res.append([out, lse] + grads)
    for a, b_ in zip(*res):
        assert torch.isfinite(b_).all() and torch.equal(a, b_)



@pytest.mark.parametrize('M,N,K,bias', [(512, 256, 128, False), (2048, 768, 768, False), (1280, 512, 192, True)])
def test_gemm_rowdot_epilogue(ops, M, N, K, bias):
    """PB_GEMM_ROWDOT (round 5): C = A B^T (+ bias) as usual, and out[n / 64][m] = sum over the 64-column group of bf16(C[m][.]) * aux[m][.] --
    the delta = rowsum(dO * O) per head of the attention backward, taken from the epilogue registers of the GEMM that makes dO. Against
    fp64 on the ROUNDED C (what a pass over the stored tensor sees), one and several work items per workgroup; refused off the whole-tile form."""
    from pianobart_amd._lib import PBError
    g = torch.Generator(device='cuda').manual_seed(M + N)
    A = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    B = torch.randn(N, K, device='cuda', generator=g).to(torch.bfloat16)
    aux = torch.randn(M, N, device='cuda', generator=g).to(torch.bfloat16)
    b = torch.randn(N, device='cuda', generator=g) if bias else None
    C = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
    ld = M + 64
    out = torch.full((N // 64, ld), float('nan'), device='cuda')
    ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, bias=b, rowdot=(aux, out, ld))
    ref = A.double() @ B.double().t() + (b.double() if bias else 0)
    assert float((C.double() - ref).abs().max() / ref.abs().max()) < 6e-3
    C2 = torch.empty_like(C)
    ops.gemm(A, B, C2, M=M, N=N, K=K, dtype=ops.PB_BF16, bias=b)
    assert torch.equal(C, C2)                                             # the same C as without the extra output
    want = (C.double() * aux.double()).reshape(M, N // 64, 64).sum(-1).t()
    got = out[:, :M].double()
    assert torch.isfinite(got).all() and torch.isnan(out[:, M:]).all()
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max()) + 1e-4
    with pytest.raises(PBError):                                          # a ragged row count is not the whole-tile form
        ops.gemm(A[:M - 8], B, C[:M - 8], M=M - 8, N=N, K=K, dtype=ops.PB_BF16, rowdot=(aux, out, ld))
    # ... and every route to the generic kernel (which has no such epilogue) refuses instead of leaving `out` unwritten:
    out.fill_(float('nan'))
    with pytest.raises(PBError):                                          # PB_GEMM_FORCE_V1 (also reachable through PB_GEMM_FLAGS=16)
        ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, rowdot=(aux, out, ld), force_v1=True)
    with pytest.raises(PBError):                                          # an operand off 16 bytes: the tiled kernels decline it
        Am = torch.empty(M * K + 8, device='cuda', dtype=torch.bfloat16)[1:M * K + 1].view(M, K)
        ops.gemm(Am, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, rowdot=(aux, out, ld))
    with pytest.raises(PBError):                                          # K no multiple of 64
        ops.gemm(A[:, :K - 32].contiguous(), B[:, :K - 32].contiguous(), C, M=M, N=N, K=K - 32, dtype=ops.PB_BF16, rowdot=(aux, out, ld))
    with pytest.raises(PBError):                                          # f32 operands
        ops.gemm(A.float(), B.float(), C.float(), M=M, N=N, K=K, dtype=ops.PB_F32, rowdot=(aux.float(), out, ld))
    torch.cuda.synchronize()
    assert torch.isnan(out).all()                                         # none of them launched anything that touched it


@pytest.mark.parametrize('M,N,K', [(1024, 768, 256), (2048 + 136, 512 + 64, 192), (4096, 3072, 768)])
def test_gemm_row_staged_epilogues_store_the_register_path_bits(ops, M, N, K):
    """The 256 x 256 kernel's interior tiles store through the row staging (8 rows x 128 contiguous bytes per instruction, through a wave-private
    LDS image); PB_GEMM_REG_EPILOGUE (256) is the path straight from the MFMA register layout. Same bits, for the store-only epilogue, the f32
    split-K slabs of the weight-gradient layout and -- against edge tiles, which always take the register path -- a ragged problem; the GELU pair,
    the read-modify-write epilogues and the row sums (no register twin left) are pinned by their own tests above and by tools/gemm_race_screen.py."""
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N)
    rn = lambda *s_: torch.randn(*s_, device='cuda', generator=g).to(torch.bfloat16)
    A, B = rn(M, K), rn(N, K)
    bias = torch.randn(N, device='cuda', generator=g)
    out = []
    for dbg in (0, 256):
        C = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
        ops.gemm(A, B, C, M=M, N=N, K=K, dtype=ops.PB_BF16, bias=bias, tile256=True, dbg=dbg)
        assert torch.isfinite(C).all()
        out.append(C)
    assert torch.equal(out[0], out[1])
    ref = A.double() @ B.double().t() + bias.double()
    assert float((out[0].double() - ref).abs().max() / ref.abs().max()) < 6e-3
    # weight-gradient layout: G (Mw x Nw) = dY^T X over T rows, split-K into f32 slabs
    T, Mw, Nw, ns = 4096, 768, 512, 4
    dy, x = rn(T, Mw), rn(T, Nw)
    slabs = torch.empty(ns * Mw * Nw, device='cuda')
    res = []
    for dbg in (0, 256):
        G = torch.zeros(Mw, Nw, device='cuda')
        ops.gemm(dy, x, G, M=Mw, N=Nw, K=T, dtype=ops.PB_BF16, a_kc=False, b_kc=False, lda=Mw, ldb=Nw, ldc=Nw, c_f32=True, splitk=ns, slabs=slabs, tile256=True, dbg=dbg)
        res.append(G)
    assert torch.equal(res[0], res[1])
    refg = dy.double().t() @ x.double()
    assert float((res[0].double() - refg).abs().max() / refg.abs().max()) < 2e-3


def test_one_pass_backward_takes_delta_rows(ops):
    """pb_flash_bwd1 with delta_rows = [H][B * S] row sums made elsewhere (the GEMM epilogue) gives the gradients of the call that computes
    delta itself, bit for bit when the table holds the same numbers."""
    hd, B, H, S = 64, 2, 3, 320
    d = H * hd
    g = torch.Generator(device='cuda').manual_seed(5)
    buf = (torch.randn(B, S, 3 * d, device='cuda', generator=g) * 1.2).to(torch.bfloat16)
    ql, kl, vl = (buf, 0, 3 * d, S * 3 * d), (buf, d, 3 * d, S * 3 * d), (buf, 2 * d, 3 * d, S * 3 * d)
    km = (torch.rand(B, S, device='cuda', generator=g) > 0.2).float()
    kmax = torch.empty(B, dtype=torch.int32, device='cuda'); ops.key_extent(km, kmax)
    out = torch.empty(B, S, d, device='cuda', dtype=torch.bfloat16); lse = torch.empty(B, H, S, device='cuda')
    ops.flash_fwd(ql, kl, vl, (out, 0, d, S * d), lse, km, B, H, S, S, hd, hd ** -0.5, False, kmax=kmax)
    dout = torch.randn(B, S, d, device='cuda', generator=g).to(torch.bfloat16)
    res = []
    for use_rows in (False, True):
        delta = torch.zeros(B, H, S, device='cuda')
        dbuf = torch.full((B, S, 3 * d), float('nan'), device='cuda', dtype=torch.bfloat16)
        dq, dk, dv = (dbuf, 0, 3 * d, S * 3 * d), (dbuf, d, 3 * d, S * 3 * d), (dbuf, 2 * d, 3 * d, S * 3 * d)
        rows = None
        if use_rows:                                                      # [H][B * S] from the first call's (B, H, S) table
            rows = res[0][1].permute(1, 0, 2).reshape(H, B * S).contiguous()
        ops.flash_bwd1(ql, kl, vl, (out, 0, d, S * d), dout, lse, km, dq, dk, dv, delta, B, H, S, S, hd, hd ** -0.5, False, kmax=kmax, delta_rows=rows)
        res.append((dbuf.clone(), delta.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.isfinite(res[0][0]).all()
    assert float(res[1][1].abs().max()) == 0.0                            # the second call did not run the delta pass


def test_one_pass_backward_masked_keys_with_a_very_negative_log_sum_exp(ops):
    """the one-pass kernel hides a masked key by zeroing its K fragments, so that key's score is exactly -lse; for a row whose scores all
    lie below about -88 (log-sum-exp < -88) exp2(-lse log2e) overflowed to inf and inf x 0 made NaNs in dQ. Rows like that now send the waves that
    hold a masked key through the clamped compare path. q = +a, k = -a (+ noise): every score ~ -110; 25 % of the keys masked; vs fp64."""
    hd, B, H, S = 64, 2, 2, 384
    d = H * hd
    g = torch.Generator(device='cuda').manual_seed(11)
    a = 3.7
    q = (a + 0.1 * torch.randn(B, S, d, device='cuda', generator=g)).to(torch.bfloat16)
    k = (-a + 0.1 * torch.randn(B, S, d, device='cuda', generator=g)).to(torch.bfloat16)
    v = torch.randn(B, S, d, device='cuda', generator=g).to(torch.bfloat16)
    buf = torch.cat([q, k, v], -1).contiguous()
    ql, kl, vl = (buf, 0, 3 * d, S * 3 * d), (buf, d, 3 * d, S * 3 * d), (buf, 2 * d, 3 * d, S * 3 * d)
    km = (torch.rand(B, S, device='cuda', generator=g) > 0.25).float()
    kmax = torch.empty(B, dtype=torch.int32, device='cuda'); ops.key_extent(km, kmax)
    out = torch.empty(B, S, d, device='cuda', dtype=torch.bfloat16); lse = torch.empty(B, H, S, device='cuda')
    scale = hd ** -0.5
    ops.flash_fwd(ql, kl, vl, (out, 0, d, S * d), lse, km, B, H, S, S, hd, scale, False, kmax=kmax)
    assert float(lse.max()) < -95.0                                       # the regime: exp2(-lse log2e) = 2^137 and more
    dout = torch.randn(B, S, d, device='cuda', generator=g).to(torch.bfloat16)
    delta = torch.empty(B, H, S, device='cuda')
    dbuf = torch.full((B, S, 3 * d), float('nan'), device='cuda', dtype=torch.bfloat16)
    ops.flash_bwd1(ql, kl, vl, (out, 0, d, S * d), dout, lse, km, (dbuf, 0, 3 * d, S * 3 * d), (dbuf, d, 3 * d, S * 3 * d), (dbuf, 2 * d, 3 * d, S * 3 * d),
                   delta, B, H, S, S, hd, scale, False, kmax=kmax)
    assert torch.isfinite(dbuf).all()
    bd = buf.double().requires_grad_(True)
    q4, k4, v4 = (bd[..., i * d:(i + 1) * d].reshape(B, S, H, hd).permute(0, 2, 1, 3) for i in range(3))
    vis = (km != 0)[:, None, None, :].expand(B, H, S, S)
    p = torch.softmax((q4 @ k4.transpose(2, 3) * scale).masked_fill(~vis, float('-inf')), -1)
    ((p @ v4).permute(0, 2, 1, 3).reshape(B, S, d)).backward(dout.double())
    err = float((dbuf.double() - bd.grad).abs().max() / bd.grad.abs().max())
    assert err < 8e-2, err             # measured 4.5e-2: at |score| ~ 110 the bf16 rounding of the prescaled K alone moves an exponent by ~0.2
    assert float(dbuf[..., d:].double()[(km == 0)[..., None].expand(B, S, 2 * d)].abs().max()) == 0.0     # masked keys: zero dK / dV rows


def test_split_bf16_planes(ops):
    """pb_split_bf16 (round 6): x -> hi = bf16(x) (round to nearest even, what torch's cast gives), lo = bf16(x - hi); hi + lo restores x to 2^-17 relative -- the two planes the
    bf16x3 embedding-table gradient feeds to its pair of one-hot GEMMs."""
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(1024, 264, device='cuda', generator=g) * torch.logspace(-6, 3, 264, device='cuda')
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-38, 65504.0], device='cuda')
    hi = torch.full(x.shape, float('nan'), device='cuda', dtype=torch.bfloat16); lo = torch.full_like(hi, float('nan'))
    ops.split_bf16(x, hi, lo)
    assert torch.equal(hi, x.to(torch.bfloat16))
    assert torch.equal(lo, (x - hi.float()).to(torch.bfloat16))
    err = ((hi.double() + lo.double()) - x.double()).abs() / x.double().abs().clamp(min=1e-30)
    assert float(err[x.abs() > 1e-30].max()) < 2.0 ** -16


@pytest.mark.parametrize('M,N,K,akc,bkc', [(512, 384, 256, True, True), (300, 136, 96, True, True), (2048, 768, 768, True, False), (768, 3072, 4096, False, False),
                                           (264, 72, 40, False, True), (1024, 1280, 768, True, True)])
def test_gemm_f32x3_against_fp64(ops, M, N, K, akc, bkc):
    """PB_F32X3 (round 6): f32 operands as split-bf16 triples on the bf16 kernels, f32 C. Against fp64: ~1e-5 of the output scale (the exact-f32
    kernel gives ~1e-6, the bf16 one ~4e-3), for both operand layouts, ragged sizes (K segments are zero-padded to 64), bias / alpha / accumulate,
    the GELU pair and the multiply by a stored derivative (f32 aux tensors), column sums, and split-K into slabs."""
    g = torch.Generator(device='cuda').manual_seed(M + 3 * N + K)
    rn = lambda *s_: torch.randn(*s_, device='cuda', generator=g)
    A = rn(M, K) if akc else rn(K, M)
    B = rn(N, K) if bkc else rn(K, N)
    bias = rn(N)
    ref = (A.double() if akc else A.double().t()) @ (B.double().t() if bkc else B.double())
    scale = float(ref.abs().max())
    X3 = ops.PB_F32X3
    C = torch.full((M, N), float('nan'), device='cuda')
    ops.gemm(A, B, C, M=M, N=N, K=K, dtype=X3, a_kc=akc, b_kc=bkc, c_f32=True)
    e_plain = float((C.double() - ref).abs().max()) / scale
    Cb = torch.empty_like(C)
    ops.gemm(A.to(torch.bfloat16), B.to(torch.bfloat16), Cb, M=M, N=N, K=K, dtype=ops.PB_BF16, a_kc=akc, b_kc=bkc, c_f32=True)
    e_bf16 = float((Cb.double() - ref).abs().max()) / scale
    print('f32x3 %dx%dx%d: %.2e of max (bf16 operands: %.2e)' % (M, N, K, e_plain, e_bf16))
    assert e_plain < 2e-5 and e_plain * 50 < e_bf16
    C2 = torch.ones(M, N, device='cuda')
    ops.gemm(A, B, C2, M=M, N=N, K=K, dtype=X3, a_kc=akc, b_kc=bkc, c_f32=True, bias=bias, alpha=0.5, accum=True)
    want = 1.0 + 0.5 * ref + bias.double()
    assert float((C2.double() - want).abs().max()) / scale < 2e-5
    U = torch.empty(M, N, device='cuda'); D = torch.empty(M, N, device='cuda')
    ops.gemm(A, B, U, M=M, N=N, K=K, dtype=X3, a_kc=akc, b_kc=bkc, c_f32=True, bias=bias, alpha=0.05, gelu_aux_out=D)
    u = 0.05 * ref + bias.double()
    cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
    assert float((U.double() - u * cdf).abs().max()) < 1e-5 * max(1.0, float(u.abs().max()))
    assert float((D.double() - (cdf + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5)).abs().max()) < 2e-4          # gelu_grad_f: __expf, as in the exact-f32 epilogue
    G = torch.empty(M, N, device='cuda')
    cs = torch.zeros(N, device='cuda')
    ws = torch.empty(int(ops.LIB.query('pb_gemm_colsum_ws_floats', M, N)), device='cuda')
    ops.gemm(A, B, G, M=M, N=N, K=K, dtype=X3, a_kc=akc, b_kc=bkc, c_f32=True, gelu_grad_aux_in=D, colsum_out=cs, colsum_ws=ws)
    wantg = ref * D.double()
    assert float((G.double() - wantg).abs().max()) / scale < 2e-5
    assert float((cs.double() - wantg.sum(0)).abs().max()) < 1e-4 * float(wantg.sum(0).abs().max() + scale)
    if K % 64 == 0 and K >= 256:
        slabs = torch.empty(4 * M * N, device='cuda')
        S4 = torch.empty(M, N, device='cuda')
        ops.gemm(A, B, S4, M=M, N=N, K=K, dtype=X3, a_kc=akc, b_kc=bkc, c_f32=True, splitk=4, slabs=slabs)
        assert float((S4.double() - ref).abs().max()) / scale < 2e-5


def test_gemm_f32x3_batched_strided_like_the_unfused_attention(ops):
    """The unfused attention products of the parity instantiations address heads by stride inside (T, 3d) rows: Q K^T (K = head_dim, NT) and
    P V (B operand row-contiguous) in PB_F32X3 against fp64."""
    g = torch.Generator(device='cuda').manual_seed(5)
    B_, H, S, hd = 2, 4, 96, 32
    d = H * hd
    qkv = torch.randn(B_ * S, 3 * d, device='cuda', generator=g)
    scores = torch.empty(B_, H, S, S, device='cuda')
    ops.gemm(qkv, qkv, scores, M=S, N=S, K=hd, dtype=ops.PB_F32X3, lda=3 * d, ldb=3 * d, ldc=S, c_f32=True, nb1=B_, nb2=H,
             sA=(S * 3 * d, hd), sB=(S * 3 * d, hd), sC=(H * S * S, S * S), a_off=0, b_off=d)
    q = qkv[:, :d].view(B_, S, H, hd).permute(0, 2, 1, 3).double()
    k = qkv[:, d:2 * d].view(B_, S, H, hd).permute(0, 2, 1, 3).double()
    v = qkv[:, 2 * d:].view(B_, S, H, hd).permute(0, 2, 1, 3).double()
    ref = q @ k.transpose(-1, -2)
    assert float((scores.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    P = torch.softmax(scores, -1).contiguous()
    out = torch.zeros(B_ * S, d, device='cuda')
    ops.gemm(P, qkv, out, M=S, N=hd, K=S, dtype=ops.PB_F32X3, b_kc=False, lda=S, ldb=3 * d, ldc=d, c_f32=True, nb1=B_, nb2=H,
             sA=(H * S * S, S * S), sB=(S * 3 * d, hd), sC=(S * d, hd), b_off=2 * d)
    want = (P.double() @ v).permute(0, 2, 1, 3).reshape(B_ * S, d)
    assert float((out.double() - want).abs().max()) / float(want.abs().max()) < 2e-5


@pytest.mark.parametrize('hd', [32, 64, 128])
@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('S', [64, 200, 136, 328])          # 328: at 256 rows and more the operands are cut into bf16 planes by a prepass
def test_flash_attention_x3_fwd_bwd(ops, hd, causal, S):
    """Fused attention of the bf16x3 instantiation (f32 tensors, split-bf16 triples on the bf16 MFMA, f32 softmax) vs an fp64 reference: key-padding
    masks, causal, ragged S, a query without a visible key (zero row). ~1e-5 of the output scale, where the bf16 kernels are at 1e-2."""
    g = torch.Generator(device='cuda').manual_seed(hd + S)
    B, H = 2, 3
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, device='cuda', generator=g) * 1.5
    km = (torch.rand(B, S, device='cuda', generator=g) > 0.25).float()
    km[0, 0] = 0
    km[1, S // 2:] = 0
    out = torch.full((B, S, d), float('nan'), device='cuda')
    lse = torch.empty(B, H, S, device='cuda')
    scale = hd ** -0.5
    sl = lambda off: (qkv, off, 3 * d, S * 3 * d)
    ops.flash_fwd_x3(sl(0), sl(d), sl(2 * d), (out, 0, d, S * d), lse, km, B, H, S, S, hd, scale, causal)
    qd = qkv.double().requires_grad_(True)
    q = qd[..., :d].reshape(B, S, H, hd).permute(0, 2, 1, 3)
    k = qd[..., d:2 * d].reshape(B, S, H, hd).permute(0, 2, 1, 3)
    v = qd[..., 2 * d:].reshape(B, S, H, hd).permute(0, 2, 1, 3)
    vis = (km != 0)[:, None, None, :].expand(B, H, S, S)
    if causal:
        vis = vis & torch.ones(S, S, dtype=torch.bool, device='cuda').tril()
    s = (q @ k.transpose(2, 3) * scale).masked_fill(~vis, float('-inf'))
    p = torch.nan_to_num(torch.where(vis.any(-1, keepdim=True), torch.softmax(s, -1), torch.zeros_like(s)), nan=0.0)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    e_out = float((out.double() - ref).abs().max() / ref.abs().max())
    has = vis.any(-1)
    e_lse = float((lse.double() - torch.logsumexp(s, -1))[has].abs().max())
    assert torch.isinf(lse[~has]).all() and bool((out.reshape(B, S, H, hd).permute(0, 2, 1, 3)[~has] == 0).all())
    dout = torch.randn(B, S, d, device='cuda', generator=g)
    ref.backward(dout.double())
    dqkv = torch.full((B, S, 3 * d), float('nan'), device='cuda')
    delta = torch.empty(B, H, S, device='cuda')
    dsl = lambda off: (dqkv, off, 3 * d, S * 3 * d)
    ops.flash_bwd_x3(sl(0), sl(d), sl(2 * d), (out, 0, d, S * d), dout, lse, km, dsl(0), dsl(d), dsl(2 * d), delta, B, H, S, S, hd, scale, causal)
    e_grad = float((dqkv.double() - qd.grad).abs().max() / qd.grad.abs().max())
    print('flash x3 hd=%d S=%d causal=%d: out %.2e  lse %.2e  grads %.2e' % (hd, S, causal, e_out, e_lse, e_grad))
    assert e_out < 3e-5 and e_lse < 1e-4 and e_grad < 5e-5
    # key extents (1 + last visible key per batch row): the tiles behind them are skipped, the results do not change by a bit
    kmax = torch.empty(B, dtype=torch.int32, device='cuda')
    ops.key_extent(km, kmax)
    out2 = torch.full_like(out, float('nan')); lse2 = torch.empty_like(lse); dqkv2 = torch.full_like(dqkv, float('nan'))
    d2 = lambda off: (dqkv2, off, 3 * d, S * 3 * d)
    ops.flash_fwd_x3(sl(0), sl(d), sl(2 * d), (out2, 0, d, S * d), lse2, km, B, H, S, S, hd, scale, causal, kmax=kmax)
    ops.flash_bwd_x3(sl(0), sl(d), sl(2 * d), (out2, 0, d, S * d), dout, lse2, km, d2(0), d2(d), d2(2 * d), delta, B, H, S, S, hd, scale, causal, kmax=kmax)
    assert torch.equal(out, out2) and torch.equal(lse, lse2) and torch.equal(dqkv, dqkv2)


@pytest.mark.parametrize('hd', [32, 64])
@pytest.mark.parametrize('causal', [False, True])
def test_flash_attention_x3_two_tiles_per_wave_changes_no_bit(ops, hd, causal):
    """The query-stationary split-bf16 kernels (forward, dQ) give a wave TWO 16-query tiles -- blocks of 128 queries -- once the grid is large enough
    (pb_flash_x3.hip fx_nt: head_dim <= 64 and >= 1024 workgroups; every staged K / V fragment is then used twice). A query's arithmetic does not depend on the
    blocking: the whole batch (large grid, two tiles per wave) equals its batch rows run one at a time (small grids, one tile per wave) bit for bit -- ragged
    key masks, causal (the 128-query block visits a key tile more, fully masked for its first half)."""
    g = torch.Generator(device='cuda').manual_seed(7 + hd)
    B, H, S = 8, 16, 1000                       # 8 x 16 x ceil(1000 / 128) = 1024 workgroups of 128 queries
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, device='cuda', generator=g)
    km = (torch.rand(B, S, device='cuda', generator=g) > 0.2).float()
    km[1, S // 3:] = 0
    dout = torch.randn(B, S, d, device='cuda', generator=g)
    scale = hd ** -0.5

    def run(qkv_, km_, dout_, Bn):
        out = torch.full((Bn, S, d), float('nan'), device='cuda'); lse = torch.empty(Bn, H, S, device='cuda')
        dqkv = torch.full((Bn, S, 3 * d), float('nan'), device='cuda'); delta = torch.empty(Bn, H, S, device='cuda')
        sl = lambda off: (qkv_, off, 3 * d, S * 3 * d)
        dsl = lambda off: (dqkv, off, 3 * d, S * 3 * d)
        ops.flash_fwd_x3(sl(0), sl(d), sl(2 * d), (out, 0, d, S * d), lse, km_, Bn, H, S, S, hd, scale, causal)
        ops.flash_bwd_x3(sl(0), sl(d), sl(2 * d), (out, 0, d, S * d), dout_, lse, km_, dsl(0), dsl(d), dsl(2 * d), delta, Bn, H, S, S, hd, scale, causal)
        return out, lse, dqkv
    out, lse, dqkv = run(qkv, km, dout, B)
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    for b in range(B):
        o1, l1, g1 = run(qkv[b:b + 1].contiguous(), km[b:b + 1].contiguous(), dout[b:b + 1].contiguous(), 1)
        assert torch.equal(o1[0], out[b]) and torch.equal(l1[0], lse[b]) and torch.equal(g1[0], dqkv[b]), b
