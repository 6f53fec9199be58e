"""CPU, world_size 2 over gloo: the data-parallel scheme of pianobart_amd.parallel.

(1) GradReducer: buckets announced by the engine hook are all-reduced (SUM) and waited for.
(2) The DP loss semantics: per-rank gradients of  sum_local(ce*m) * w_i / (sum_w * M_i^global), summed over
    ranks, equal the single-process gradient of the reference's global masked mean (pretrain.py:117,185-189).
    The oracle stands in for the model arithmetic here (there is no CPU product path); what is under test
    is the reduction scheme itself."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeEngine:
    def __init__(self, n):
        self.G32 = torch.zeros(n)
        self.grad_hook = None


class _TorchXfer:
    """Stand-in for the three elementwise HIP kernels of the bf16 exchange (cast, f32 row sum, cast back): there is no CPU product
    path, and what this test is about is the exchange scheme around them."""
    to_bf16 = staticmethod(lambda src, dst: dst.copy_(src.to(torch.bfloat16)))
    sum_rows = staticmethod(lambda src, dst, rows: dst.copy_(src.view(rows, -1).float().sum(0).to(torch.bfloat16)))
    to_f32 = staticmethod(lambda src, dst: dst.copy_(src.float()))


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pianobart_amd.parallel import GradReducer
        # ---- (1) bucket hook
        # ---- (0) bf16 exchange: all-to-all of bf16 chunks, f32 accumulation at the owner, all-gather; ragged bucket sizes
        eng = _FakeEngine(1000)
        red = GradReducer(eng, world, mode='bf16', xfer=_TorchXfer)
        base = torch.linspace(-3.0, 7.0, 1000) ** 3
        eng.G32[:] = base * (rank + 1) * 1.01
        eng.grad_hook(600, 1000); eng.grad_hook(0, 597); eng.grad_hook(597, 600)
        got_ranges = list(red.ranges)
        red.all_reduce_grads()
        bf = lambda t: t.to(torch.bfloat16).float()
        want = bf(bf(base * 1.01) + bf(base * 2.02))
        ok0 = torch.equal(eng.G32, want) and got_ranges == [(600, 1000), (0, 597), (597, 600)] and red.ranges == []
        both = [torch.empty_like(eng.G32) for _ in range(world)]
        dist.all_gather(both, eng.G32)
        ok0 = ok0 and torch.equal(both[0], both[1])                       # every rank holds bit-identical gradients
        ok0 = ok0 and float((want - base * 3.03).abs().max() / (base * 3.03).abs().max()) < 8e-3
        eng = _FakeEngine(1000)
        red = GradReducer(eng, world, mode='f32')
        eng.G32[:] = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        eng.grad_hook(600, 1000)
        eng.grad_hook(0, 600)
        eng.grad_hook(5, 5)               # empty bucket is ignored
        red.all_reduce_grads()
        ok1 = ok0 and torch.allclose(eng.G32, torch.arange(1000, dtype=torch.float32) * 3) and red.pending == []
        counts = torch.tensor([1.0 + rank] * 8)
        red.reduce_counts(counts)
        ok1 = ok1 and torch.allclose(counts, torch.full((8,), 3.0))
        # ---- (2) loss semantics with the oracle as the model
        from oracle import pianobart_oracle as O
        from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
        e2w, w2e = load_vocab()
        cfg = O.BartConfig(max_position_embeddings=32, d_model=32, encoder_layers=1, decoder_layers=1, encoder_ffn_dim=64,
                           decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2, dropout=0.0)
        m = O.PianoBartLM(O.PianoBart(cfg, e2w, w2e)).train()
        randomize_params(m, 5)
        enc, dec, lm, em, dm, tgt = synth_octuple_batch(4, 32, seed=3)
        lm[0] = 0; lm[0, :9] = 1                         # very different mask counts per rank
        w = torch.tensor(O.loss_weights(e2w), dtype=torch.float32)

        def grads_of(loss):
            m.zero_grad(); loss.backward()
            return torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])

        total, *_ = O.pretrain_loss(m(enc, dec, em, dm), tgt, lm, e2w)
        g_ref = grads_of(total)
        sl = slice(rank * 2, rank * 2 + 2)
        counts = lm[sl].reshape(-1, 8).sum(0)
        red.reduce_counts(counts)                        # global M_i
        y = m(enc[sl], dec[sl], em[sl], dm[sl])
        local = 0
        for i in range(8):
            ce = torch.nn.functional.cross_entropy(y[i].permute(0, 2, 1), tgt[sl][..., i], reduction='none')
            local = local + (ce * lm[sl][..., i]).sum() * w[i] / (w.sum() * counts[i])
        g = grads_of(local)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        ok2 = float((g - g_ref).abs().max() / g_ref.abs().max()) < 1e-5
        q.put((rank, bool(ok1), bool(ok2)))
    finally:
        dist.destroy_process_group()


def test_dp_reducer_and_loss_semantics_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]
