"""GPU, world size 1 over RCCL: the data-parallel step as the engine really issues it (the 8-GPU node is the driver's).

With the GradReducer installed the engine (a) announces flat gradient ranges as their kernels are enqueued -- they must tile
[0, n_total) exactly once -- (b) launches its backward GEMMs as ordinary grids instead of persistent ones, and (c) hands every
bucket to the exchange from the stream that produced it. At world size 1 the exchange is the identity (f32 mode) or one bf16
rounding (bf16 mode), so the step must reproduce the plain step bit for bit / to bf16 rounding."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch

pytestmark = pytest.mark.gpu
E2W, W2E = load_vocab()


@pytest.fixture(scope='module')
def pg():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(29600 + os.getpid() % 300)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield dist
    dist.destroy_process_group()


def _engine(precision):
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    cfg = BartConfig(max_position_embeddings=256, d_model=256, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                     encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.1)
    m = PianoBartLM(PianoBart(cfg, E2W, W2E, precision=precision))
    randomize_params(m, 31)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    return m, eng


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_reducer_step_equals_plain_step_and_ranges_tile_the_buffer(pg, precision, mode):
    from pianobart_amd import ops
    from pianobart_amd.parallel import GradReducer
    m, eng = _engine(precision)
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(4, 256, seed=3)]
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    plain = []
    for it in range(2):
        eng._seed = 500 + it
        s = eng.loss_and_grads(*args, train=True)
        torch.cuda.synchronize()
        plain.append((eng.G32.clone(), s.clone()))
    red = GradReducer(eng, 1, mode=mode)
    seen = []
    inner = red._on_ready
    eng.grad_hook = lambda lo, hi: (seen.append((lo, hi)), inner(lo, hi))[1]
    try:
        for it in range(2):
            seen.clear()
            eng._seed = 500 + it
            s = eng.loss_and_grads(*args, train=True, count_hook=red.reduce_counts)
            red.all_reduce_grads()
            torch.cuda.synchronize()
            # (a) every flat gradient element is announced exactly once
            cover = np.zeros(eng.n_total, dtype=np.int32)
            for lo, hi in seen:
                assert 0 <= lo <= hi <= eng.n_total
                cover[lo:hi] += 1
            assert (cover == 1).all(), 'gradient ranges announced %s times' % sorted(set(cover.tolist()))
            g0, s0 = plain[it]
            assert torch.equal(s0, s)
            if mode == 'f32':
                # (b)+(c): ordinary grids + exchange from the producing stream give the plain step's gradients, bit for bit
                atomic = ('emb', 'lin.w', 'enc.pos', 'dec.pos') if precision == 'fp32' else ()       # f32 route: atomics, order varies
                for name, sl in eng.slots.items():
                    a, b = g0[sl.off:sl.off + sl.numel], eng.G32[sl.off:sl.off + sl.numel]
                    if name in atomic:
                        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), name
                    else:
                        assert torch.equal(a, b), (name, float((a - b).abs().max()))
            else:
                # one bf16 rounding of the input, an f32 sum of one term, one more (idempotent) rounding
                want = g0.to(torch.bfloat16).float()
                if precision == 'bf16':
                    assert torch.equal(want, eng.G32), float((want - eng.G32).abs().max())
                else:
                    # exact-f32 engine: its embedding-table gradients are f32 atomics (order varies), so a bf16 rounding may flip
                    assert float((want - eng.G32).abs().max() / want.abs().max()) < 8e-3
    finally:
        red.close()


def test_sum_rows_bf16_kernel(pg):
    from pianobart_amd import ops
    g = torch.Generator(device='cuda').manual_seed(2)
    for rows, n in ((8, 4096), (2, 8), (5, 100000)):
        src = torch.randn(rows, n, device='cuda', generator=g).to(torch.bfloat16)
        dst = torch.empty(n, device='cuda', dtype=torch.bfloat16)
        ops.sum_rows_bf16(src, dst, rows)
        acc = torch.zeros(n, device='cuda')
        for r in range(rows):
            acc += src[r].float()
        assert torch.equal(dst, acc.to(torch.bfloat16))


def test_finetune_trainer_data_parallel_path_equals_plain(pg):
    """FinetuneTrainer with the data-parallel path installed at world size 1 (bucket exchange of the backbone gradients, all-reduce
    of the head optimizer's flat gradient buffer, 1 / world scaling, --weight regulariser) takes the same steps as the plain trainer."""
    from torch.utils.data import DataLoader
    from pianobart_amd import model as M
    from pianobart_amd.finetune import FinetuneDataset, FinetuneTrainer
    S_, D_ = 64, 128
    kw = dict(max_position_embeddings=S_, d_model=D_, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=256, decoder_ffn_dim=256,
              encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.0)
    X = synth_octuple_batch(8, S_, seed=3)[0].numpy()
    y = np.random.default_rng(0).integers(0, 8, size=(8,))
    res = []
    for dp in (None, True):
        torch.manual_seed(0)
        pb = M.PianoBart(M.BartConfig(**kw), E2W, W2E, precision='fp32')
        randomize_params(pb, 3)
        mk = lambda: DataLoader(FinetuneDataset(X, y), batch_size=4)
        tr = FinetuneTrainer(pb, mk(), mk(), mk(), lr=1e-3, class_num=8, hs=D_, testset_shape=y.shape, cpu=False, cuda_devices=[0], SeqClass=True,
                             weight=0.01, data_parallel=dp)
        for m_ in tr.model.modules():
            if isinstance(m_, torch.nn.Dropout):
                m_.p = 0.0
        randomize_params(tr.model.classifier, 5); randomize_params(tr.model.attention, 6)
        losses = [tr.train()[0] for _ in range(3)]
        res.append((losses, tr.engine.P32.clone(), tr.head_optim.P.clone()))
        tr.reducer.close() if getattr(tr, 'reducer', None) is not None else None
    (l0, p0, h0), (l1, p1, h1) = res
    assert all(abs(a - b) < 2e-4 for a, b in zip(l0, l1)) and l0[-1] < l0[0]
    # (the default exchange is an f32 all-reduce: at world size 1 the identity; the bf16 mode would round the gradients once)
    assert float((p0 - p1).abs().max()) < 1e-4 and float((h0 - h1).abs().max()) < 1e-4


def test_exchange_is_independent_of_the_packed_shape(pg, monkeypatch):
    """Under data parallelism every rank packs ITS OWN samples: the rows a rank keeps (Te, Td, loss rows) differ from rank to rank and
    from step to step, while the exchange must see the same flat ranges, in the same order, with the same sizes on every rank (the
    collectives pair up by issue order). Two batches with different packing (and one that stays dense) through one reducer: the
    announced (lo, hi) sequences are identical, each tiles [0, n_total) once, reduce_counts moves the same 8 floats, and every step's
    gradients equal the plain step's rounded to bf16 (pretrain.py:63-65 semantics: a sum over replicas of per-rank gradients)."""
    from pianobart_amd import engine as E, ops
    from pianobart_amd.parallel import GradReducer
    m, eng = _engine('bf16')
    batches = []
    for seed, min_len in ((3, None), (11, 40), (12, 250)):
        kw = {} if min_len is None else {'min_len': min_len}
        enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(8, 256, seed=seed, **kw)]
        batches.append((ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask))
    plain, shapes = [], []
    for it, args in enumerate(batches):
        eng._seed = 900 + it
        eng.loss_and_grads(*args, train=True)
        torch.cuda.synchronize()
        plain.append(eng.G32.clone())
        shapes.append(eng.last_rows)
    T = 8 * 256
    assert len(set(shapes)) >= 2 and any(sh[0] < T for sh in shapes) and any(sh[0] == T for sh in shapes), shapes      # packed layouts and a dense one
    red = GradReducer(eng, 1, mode='bf16')
    seen, counts_seen = [], []
    inner = red._on_ready
    eng.grad_hook = lambda lo, hi: (seen.append((lo, hi)), inner(lo, hi))[1]
    hook = lambda c: (counts_seen.append(tuple(c.shape)), red.reduce_counts(c))[1]
    try:
        seqs = []
        for it, args in enumerate(batches):
            seen.clear()
            eng._seed = 900 + it
            eng.loss_and_grads(*args, train=True, count_hook=hook)
            red.all_reduce_grads()
            torch.cuda.synchronize()
            assert eng.last_rows == shapes[it]
            seqs.append(list(seen))
            cover = np.zeros(eng.n_total, dtype=np.int32)
            for lo, hi in seen:
                cover[lo:hi] += 1
            assert (cover == 1).all()
            want = plain[it].to(torch.bfloat16).float()
            assert torch.equal(want, eng.G32), (it, float((want - eng.G32).abs().max()))
        assert seqs[0] == seqs[1] == seqs[2]                              # same ranges, same order, whatever was packed
        assert counts_seen == [(8,)] * 3
    finally:
        red.close()


def test_reducer_step_at_the_bench_shape(pg):
    """configs[1] model and batch (12L / 768 / S = 1024, B = 32) with the reducer installed at world size 1: the ranges tile the flat
    buffer once and the bf16-exchange step equals the plain step rounded to bf16 (packed rows, ordinary-grid backward GEMMs, the
    communication stream)."""
    from pianobart_amd import ops
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from pianobart_amd.parallel import GradReducer
    cfg = BartConfig(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
                     encoder_attention_heads=12, decoder_attention_heads=12, dropout=0.1)
    m = PianoBartLM(PianoBart(cfg, E2W, W2E, precision='bf16'))
    randomize_params(m, 41)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    enc, dec, loss_mask, emask, dmask, target = [t.cuda() for t in synth_octuple_batch(32, 1024, seed=1234)]
    args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
    eng._seed = 77
    s0 = eng.loss_and_grads(*args, train=True).clone()
    torch.cuda.synchronize()
    want = eng.G32.to(torch.bfloat16).float()
    Te, Td, T, Ts = eng.last_rows
    assert T == 32768 and Te < T and Td < T
    red = GradReducer(eng, 1, mode='bf16')
    seen = []
    inner = red._on_ready
    eng.grad_hook = lambda lo, hi: (seen.append((lo, hi)), inner(lo, hi))[1]
    try:
        eng._seed = 77
        s1 = eng.loss_and_grads(*args, train=True, count_hook=red.reduce_counts)
        red.all_reduce_grads()
        torch.cuda.synchronize()
        cover = np.zeros(eng.n_total, dtype=np.int32)
        for lo, hi in seen:
            cover[lo:hi] += 1
        assert (cover == 1).all()
        assert torch.equal(s0, s1)
        # the ordinary-grid backward GEMMs sum their K range in the persistent grid's order (same tiles, same splits): identical bits
        assert torch.equal(want, eng.G32), float((want - eng.G32).abs().max() / want.abs().max())
    finally:
        red.close()


def test_two_ranks_shares_of_the_global_bench_batch_announce_the_same_exchange(pg):
    """What an 8-GPU run of bench.py would hand ranks 0 and 5 (configs[2]: ONE global batch of 8 x 32 sequences from seed 1234, dealt by
    length in snake order: bench.rank_share), stepped here one after the other through the reducer at world size 1: the two ranks step
    different samples, yet they announce the SAME sequence of (lo, hi) gradient ranges -- every RCCL call of a
    step pairs up across ranks by construction -- the ranges tile the flat buffer once, and the count hook sees the same 8 slots."""
    import bench
    from pianobart_amd import ops
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from pianobart_amd.parallel import GradReducer
    cfg = BartConfig(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
                     encoder_attention_heads=12, decoder_attention_heads=12, dropout=0.1)
    m = PianoBartLM(PianoBart(cfg, E2W, W2E, precision='bf16'))
    randomize_params(m, 41)
    m = m.train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    glob = synth_octuple_batch(8 * 32, 1024, seed=1234)
    lengths = glob[3].sum(1).numpy()
    red = GradReducer(eng, 1, mode='bf16')
    inner_ready, inner_counts = red._on_ready, red.reduce_counts
    runs = {}
    try:
        for rank in (0, 5):
            idx = torch.tensor(bench.rank_share(lengths, 8, rank), dtype=torch.long)
            enc, dec, loss_mask, emask, dmask, target = [t[idx].contiguous().cuda() for t in glob]
            args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
            seen, counts = [], []
            eng.grad_hook = lambda lo, hi: (seen.append((lo, hi)), inner_ready(lo, hi))[1]
            hook = lambda c: (counts.append((tuple(c.shape), c.dtype, c.data_ptr())), inner_counts(c))[1]
            eng._seed = 77
            eng.loss_and_grads(*args, train=True, count_hook=hook)
            red.all_reduce_grads()
            torch.cuda.synchronize()
            assert torch.isfinite(eng.G32).all()
            runs[rank] = (seen, counts, eng.last_rows)
        (s0, c0, r0), (s5, c5, r5) = runs[0], runs[5]
        assert not set(bench.rank_share(lengths, 8, 0)) & set(bench.rank_share(lengths, 8, 5))      # different samples ...
        assert abs(r0[0] - r5[0]) < 0.02 * r0[0] and abs(r0[1] - r5[1]) < 0.02 * r0[1]      # ... whose packed sides are (nearly) equally long: the deal balances the ranks
        assert s0 == s5 and len(s0) > 20                                 # the same exchange schedule
        assert c0 == c5 and len(c0) == 1 and c0[0][0] == (8,)           # one all-reduce of the 8 mask counts
        cover = np.zeros(eng.n_total, dtype=np.int32)
        for lo, hi in s0:
            cover[lo:hi] += 1
        assert (cover == 1).all()
    finally:
        red.close()
