#!/usr/bin/env python3
"""bench.py -- PianoBART pre-train step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N>1, either way: (a) python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
  (the ranks check WORLD_SIZE == N), or (b) plain `python bench.py --gpus N`: the process starts that torchrun job as a child
  (launch_ranks: it never touches the GPU itself) and relays rank 0's line; fewer than N GPUs on the node -> non-zero exit, no line.

A "step" = one pass of the hot path over one resident synthetic Octuple batch: forward (train mode,
dropout 0.1 active) -> fused 8-head CE/argmax/acc -> full backward -> [RCCL gradient all-reduce] ->
clip(3.0) -> HF AdamW (+ bf16 shadow refresh). Corrupted inputs, decoder inputs, loss mask and
attention masks are already in HBM when the timed region starts (corruption excluded).
Workload at N=1 = BASELINE.json configs[1]: 12L/768d/ffn3072/12 heads, S=1024, B=32, bf16 (the synthetic batch of SURVEY 8d,
seed 1234). N>1 (configs[2], weak scaling): ONE global batch of N x 32 sequences from the same seed, dealt to the ranks by
sequence length exactly as the pre-training loop's BalancedDistributedSampler deals its global batches (rank_share below).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)


def train_flops_per_token(S, d, N, f, V=1280):
    """Algorithmic FLOPs of the reference graph (SURVEY 8d / BASELINE.md 4): 1 MAC = 2 FLOP, train = 3 x forward,
    causal decoder self-attention counted at 1/2, embedding merge counted as the reference's (T x 2048)(2048 x d) GEMMs."""
    macs = 2 * S * 2048 * d + N * S * (4 * d * d + 2 * d * f) + N * 2 * S * S * d + N * S * (8 * d * d + 2 * d * f) \
        + N * (S * S * d) + N * 2 * S * S * d + S * d * V
    return 3 * 2 * macs / S


def train_flops_live_rows(Te, Td, pairs, d, N, f, V=1280, Ts=None):
    """train_flops_per_token's graph restricted to the rows the step keeps (dead-row compaction): Te encoder-side and Td
    decoder-side rows, Ts <= Td rows on the query side of the LAST decoder layer (cross-attention block, FFN) and under the LM
    heads, pairs = (query, key) pairs covered per layer by the encoder self-, decoder self- (causal) and cross-attention (each
    pair costs 2 d MACs forward: QK^T and PV over all heads). With Te = Td = Ts = B S and pairs = (B S^2, B S^2 / 2, B S^2) this
    is train_flops_per_token x B S."""
    Ts = Td if Ts is None else Ts
    macs = (Te + Td) * 2048 * d + N * Te * (4 * d * d + 2 * d * f) + N * Td * 4 * d * d + ((N - 1) * Td + Ts) * (2 * d * d + 2 * d * f) \
        + N * Te * 2 * d * d + N * 2 * d * (pairs[0] + pairs[1] + pairs[2]) + Ts * d * V
    return 3 * 2 * macs


def rank_share(lengths, world, rank):
    """Which samples of a global batch rank `rank` steps: sorted by non-PAD length, dealt in snake order (0..W-1, W-1..0, ...), i.e. what
    pianobart_amd.data.BalancedDistributedSampler hands the ranks of the real pre-training loop. The packed step drops a sample's PAD
    rows, so a rank's step time follows its samples' lengths and the ranks meet at every gradient exchange."""
    import numpy as np
    order = np.argsort(-np.asarray(lengths), kind='stable')
    per = len(order) // world
    return [int(order[j * world + (rank if j % 2 == 0 else world - 1 - rank)]) for j in range(per)]


def synth_rank_batch(B, S, world, rank, device):
    """SURVEY 8(d)'s synthetic batch. One GPU: B sequences from seed 1234. N GPUs (weak scaling): every rank draws the same GLOBAL
    batch of N x B sequences from seed 1234 and keeps its share of it (rank_share)."""
    sys.path.insert(0, os.path.join(ROOT))
    from tests.golden_util import synth_octuple_batch
    t = synth_octuple_batch(B * world, S, 1234)
    if world > 1:
        idx = torch.tensor(rank_share(t[3].sum(1).numpy(), world, rank), dtype=torch.long)
        t = [x[idx].contiguous() for x in t]
    return [x.to(device) for x in t]


def cpu_baseline_child(cfgkw, S):
    """Runs in a child process (no GPU): the oracle (CPU restatement of the reference, kind "port") timed on this host's
    cores on a BOUNDED sample of the same workload: full train steps (forward, 8-head masked CE, backward, clip,
    HF AdamW) at the bench model shape and sequence length with B=1 (tokens/s is batch-insensitive on CPU; a B=32 step
    would take ~15 min): median of up to 3 steps, stopping early once ~100 s are spent. Checker code, timed only as the
    reported baseline."""
    from oracle import pianobart_oracle as O
    from tests.golden_util import load_vocab, synth_octuple_batch
    e2w, w2e = load_vocab()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))
    torch.set_num_threads(cores)
    m = O.PianoBartLM(O.PianoBart(O.BartConfig(**cfgkw), e2w, w2e)).train()
    enc, dec, loss_mask, emask, dmask, target = synth_octuple_batch(1, S, seed=3)
    params = [p for p in m.parameters()]
    opt_m = opt_v = None
    times = []
    t_start = time.time()
    for it in range(3):
        t0 = time.time()
        m.zero_grad()
        y = m(enc, dec, emask, dmask)
        total, *_ = O.pretrain_loss(y, target, loss_mask, e2w)
        total.backward()
        live = [p for p in params if p.grad is not None]
        grads = [p.grad for p in live]
        O.clip_grad_norm(grads, 3.0)
        if opt_m is None:
            opt_m = [torch.zeros_like(p) for p in live]; opt_v = [torch.zeros_like(p) for p in live]
        with torch.no_grad():
            O.hf_adamw_step([p.data for p in live], grads, opt_m, opt_v, step=it + 1, lr=2e-5)
        times.append(time.time() - t0)
        if time.time() - t_start + times[-1] > 100:
            break
    t = sorted(times)[len(times) // 2]
    print(json.dumps({"value": S / t, "unit": "tokens/s", "cores": cores, "kind": "port",
                      "sample": "oracle (torch fp32 restatement of the reference) full train step, same model shape, B=1, S=%d, "
                                "median of %d steps (%.1f s/step), %d threads" % (S, len(times), t, cores)}), flush=True)


def cpu_baseline(cfgkw, S, timeout=170):
    """Bounded: the child is killed after `timeout` seconds so the default bench run always finishes within minutes."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', json.dumps(cfgkw), str(S)],
                           capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {"value": None, "unit": "tokens/s", "cores": None, "kind": "port", "sample": "child failed: " + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "tokens/s", "cores": None, "kind": "port", "sample": "oracle train step did not finish within %d s on this host" % timeout}


def pmc_traffic(Ms, N, K):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r0x_gemm_fc1_pmc*.json,
    FETCH_SIZE x2 + WRITE_SIZE, collected offline: counters cannot be read from inside this process), averaged over the row counts
    the step launches it with (packed step: the encoder-side and the decoder-side count). None unless every shape has a record."""
    import glob
    recs = []
    for name in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_gemm_fc1_pmc*.json')), reverse=True):
        try:
            recs.append((os.path.basename(name), json.load(open(name))))
        except Exception:
            pass
    vals, srcs = [], []
    for M in Ms:
        hit = [(n, j) for n, j in recs if 'M=%d N=%d K=%d' % (M, N, K) in j.get('kernel', '')]     # same kernel, same shape, same tile as the live timing
        if not hit:
            return None, None
        vals.append(hit[0][1]['hbm_bytes_per_launch']); srcs.append('profiles/' + hit[0][0])
    return sum(vals) / len(vals), ', '.join(sorted(set(srcs)))


class StepProbe:
    """In-step per-kernel-family timing, live: every C-ABI op the engine issues is followed by one HIP event on the stream it
    was launched on; with the whole step on ONE stream (PB_WGRAD_STREAM=0 schedule) consecutive events bracket exactly one op
    (its launch gap included), so the per-family sums add up to the step. Runs AFTER the timed region; never part of `value`."""
    FAMILY = {'gemm': 'gemm', 'flash_fwd': 'attention', 'flash_bwd': 'attention', 'add_ln_fwd': 'rows', 'add_ln_bwd': 'rows',
              'embed_ln_fwd': 'rows', 'embed_ln_bwd': 'rows', 'colsum': 'rows', 'batch_sum': 'rows', 'onehot_build': 'rows', 'dropout': 'rows',
              'ce_fwd_bwd': 'loss', 'mask_count': 'loss', 'loss_coef': 'loss', 'grad_sqnorm': 'optimizer', 'clip_coef': 'optimizer',
              'adamw_step': 'optimizer', 'fill_f32': 'optimizer', 'cast_f32_to_bf16': 'optimizer', 'defer_flush': 'rows', 'key_extent': 'rows',
              'softmax_fwd': 'attention', 'softmax_bwd': 'attention', 'flash_fwd_packed': 'attention', 'flash_bwd_packed': 'attention',
              'flash_bwd1': 'attention', 'flash_bwd1_packed': 'attention', 'transpose_batch_bf16': 'optimizer', 'rowmap_build_sub': 'rows',
              'scatter_rows16': 'rows', 'ce_rows': 'loss', 'ids_check': 'rows', 'sum_rows_bf16': 'optimizer', 'cast_bf16_to_f32': 'optimizer',
              'rowmap_count': 'rows', 'rowmap_build': 'rows', 'gather_rows16': 'rows', 'pos_grad_packed': 'rows'}

    def __init__(self, ops, pairs=None):
        self.ops, self.saved, self.log, self.pairs = ops, {}, [], pairs          # pairs: Engine.last_pairs (packed attention work)

    def _label(self, name, a, kw):
        if name == 'gemm':
            fl = 2.0 * kw['M'] * kw['N'] * kw['K'] * kw.get('nb1', 1) * kw.get('nb2', 1)
            lay = ('N' if kw.get('a_kc', True) else 'T') + ('T' if kw.get('b_kc', True) else 'N')
            epi = ''.join(t for t, on in (('+bias', kw.get('bias') is not None), ('+gelu', kw.get('gelu_aux_out') is not None),
                                          ('*gelu\'', kw.get('gelu_grad_aux_in') is not None), ('+=', kw.get('accum', False)),
                                          ('+colsum', kw.get('colsum_out') is not None), (' splitk%d' % kw.get('splitk', 1), kw.get('splitk', 1) > 1)) if on)
            return 'gemm %s %dx%dx%d%s' % (lay, kw['M'], kw['N'], kw['K'], epi), fl
        if name in ('flash_fwd', 'flash_bwd', 'flash_bwd1'):                 # flash_bwd1*: the one-pass backward, flash_bwd*'s argument order
            off = 6 if name == 'flash_fwd' else 11
            B, H, Sq, Sk, hd = a[off:off + 5]
            causal = a[off + 6]
            fl = 4.0 * B * H * Sq * Sk * hd * (0.5 if causal else 1.0) * (1.0 if name == 'flash_fwd' else 2.5)
            return '%s B%d H%d S%dx%d hd%d%s' % (name, B, H, Sq, Sk, hd, ' causal' if causal else ''), fl
        if name in ('flash_fwd_packed', 'flash_bwd_packed', 'flash_bwd1_packed'):
            off = 5 if name == 'flash_fwd_packed' else 10
            rows, B, H, hd = a[off:off + 4]
            causal = a[off + 5]
            pairs = self.pairs[{'enc': 0, 'dec': 1, 'cross': 2}[rows.kind]]
            fl = 4.0 * H * hd * pairs * (1.0 if name == 'flash_fwd_packed' else 2.5)
            return '%s B%d H%d packed rows (%s, max %dx%d) hd%d' % (name, B, H, rows.kind, rows.Sq_max, rows.Sk_max, hd), fl
        return name, 0.0

    def __enter__(self):
        for name, fam in self.FAMILY.items():
            fn = getattr(self.ops, name)
            self.saved[name] = fn

            def wrap(*a, _fn=fn, _name=name, _fam=fam, **kw):
                _fn(*a, **kw)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                label, fl = self._label(_name, a, kw)
                self.log.append((_fam, label, fl, ev))
            setattr(self.ops, name, wrap)
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.ops, name, fn)

    def mark(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.log.append((None, None, 0.0, ev))

    def summary(self, nsteps, peak_tflops):
        fam, shape = {}, {}
        prev = None
        for f, label, fl, ev in self.log:
            if f is not None and prev is not None:
                ms = prev.elapsed_time(ev)
                a = fam.setdefault(f, [0.0, 0.0, 0]); a[0] += ms; a[1] += fl; a[2] += 1
                b = shape.setdefault(label, [0.0, 0.0, 0, f]); b[0] += ms; b[1] += fl; b[2] += 1
            prev = ev
        out = {}
        for f, (ms, fl, n) in fam.items():
            out[f] = {"ms_per_step": ms / nsteps, "launches_per_step": n / nsteps}
            if fl > 0:
                out[f]["tflops"] = fl / (ms * 1e-3) / 1e12
                out[f]["frac_of_mfma_peak"] = out[f]["tflops"] / peak_tflops
        top = sorted(shape.items(), key=lambda kv: -kv[1][0])[:12]
        table = [{"op": k, "calls_per_step": v[2] / nsteps, "avg_us": 1e3 * v[0] / v[2], "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[1] else None}
                 for k, v in top]
        return out, table, shape


def isolated_fc1(ops, eng, args, code, Tiso, dev):
    """The dominant kernel alone, back to back (20 launches), on random operands with its real epilogue, at Tiso rows. Not part of any number
    but `isolated_back_to_back_ms`; skipped under --no-probe so that a kernel trace of that run holds the step's launches only."""
    x = torch.randn(Tiso, args.hs, device=dev).to(eng.xdt); w = torch.randn(args.ffn, args.hs, device=dev).to(eng.xdt)
    bias = torch.randn(args.ffn, device=dev)
    out = torch.empty(Tiso, args.ffn, device=dev, dtype=eng.xdt); aux = torch.empty_like(out)
    for _ in range(3):
        ops.gemm(x, w, out, M=Tiso, N=args.ffn, K=args.hs, dtype=code, bias=bias, gelu_aux_out=aux)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nrep = 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(nrep):
        ops.gemm(x, w, out, M=Tiso, N=args.ffn, K=args.hs, dtype=code, bias=bias, gelu_aux_out=aux)
    e1.record()
    host_ms = (time.perf_counter() - t0) * 1e3 / nrep                # what the host needs to enqueue one launch (ctypes descriptor + call)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep, host_ms


def decode_bench(args, model, eng, dev, rank):
    """BASELINE configs[3]: prompt (1,S,8) with L_enc ~ S/2 then EOS + PAD, KV-cached decode; special tokens are made
    unsamplable so that every run generates `--steps` positions (random-init weights would stop at once)."""
    import numpy as np
    from tests.golden_util import synth_octuple_batch
    model.eval()
    with torch.no_grad():
        for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
            model.mask_lm.proj[i].bias[p0:] = -30.0
    S = args.seq
    enc = synth_octuple_batch(1, S, seed=7, min_len=S // 2)[5].to(dev)
    emask = (enc[:, :, 0] != 256).float()
    nsteps = min(args.steps, S)
    sampler = dict(T=model.SAMPLE_T, P=model.SAMPLE_P)
    np.random.seed(0)
    eng.generate(enc[:, :64].contiguous(), emask[:, :64].contiguous(), lambda r: torch.tensor([256, 128, 129, 256, 128, 32, 254, 49]))  # warm-up
    eng.generate(enc, emask, model.sample_row, max_new=16, sampler=sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.generate(enc, emask, model.sample_row, max_new=nsteps, sampler=sampler)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "generate tokens/s (KV-cached decode, B=1, S=%d)" % S, "value": nsteps / dt, "unit": "tokens/s", "n_gpus": 1,
                          "steps": nsteps, "warmup": 1, "ms_per_step": dt / nsteps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                          "config": {"workload": "decode %dL/%dd S=%d B=1, encoder + cross-K/V once (included in the time)" % (args.layers, args.hs, S)},
                          "decode_info": eng.last_decode}), flush=True)


def extra_measurements(args, model, eng, ops, step, peak, dev):
    """After the timed region, same process, same model: (a) the padded step (dead-row compaction off), (b) the step with the
    data-parallel gradient reducer installed at world size 1 (bf16 bucket exchange through RCCL, backward GEMMs as ordinary grids,
    cast / sum kernels, communication stream), (c) KV-cached decode at the same model shape (BASELINE configs[3]: B = 1, prompt of
    ~S/2 visible encoder rows, 200 generated positions, host-side nucleus sampling in the loop). Each key holds an "error" string
    instead of numbers if its leg could not run."""
    import numpy as np
    from pianobart_amd import engine as E
    from tests.golden_util import synth_octuple_batch
    out = {}

    def timed_steps(n):
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    B, S = args.batch, args.seq
    try:
        saved = E._PACK_ROWS
        E._PACK_ROWS = 0
        try:
            ms = timed_steps(10)
            fl = train_flops_per_token(S, args.hs, args.layers, args.ffn) * B * S
            out["padded_step"] = {"ms_per_step": ms, "step_mfma_frac": fl / (ms * 1e-3) / 1e12 / peak, "rows": B * S,
                                  "note": "PB_PACK_ROWS=0: every padded row computed, 10 steps"}
        finally:
            E._PACK_ROWS = saved
    except Exception as ex:                                            # noqa: BLE001 -- a failed leg must not cost the main line
        out["padded_step"] = {"error": repr(ex)[:200]}
    # (a') the same step on a batch the pre-training loop itself would build: the reference's 5-way corruption mix (pretrain.py:519-546,
    # here pb_corrupt, choices 1..5 drawn per sample from a fixed seed) of the SAME clean sequences, decoder input = shift-right, loss mask
    # and attention masks as Pretrainer.prepare_batch makes them. The headline batch is SURVEY 8(d)'s Bernoulli(0.15) mask, whose sparse loss
    # rows let the last decoder layer run on a fifth of the rows; the real mix puts a loss term on almost every decoder row.
    try:
        import random
        pb = model.pianobart
        tgt16 = step.batch[2]
        rnd = random.Random(1234)
        ch = torch.tensor([rnd.randint(1, 5) for _ in range(B)], dtype=torch.int32, device=dev)
        enc_r = torch.empty_like(tgt16)
        lm_r = torch.empty(B, S, 8, dtype=torch.float32, device=dev)
        ops.corrupt(tgt16, enc_r, lm_r, ch, None, 0.15, 0x1234ABCD, pb.pad_word_np, pb.mask_word_np, pb.n_tokens)
        dec_r = torch.empty_like(tgt16)
        ops.shift_right(tgt16, eng.sos16, dec_r, B, S)
        pad = int(pb.bar_pad_word)
        em_r, dm_r = (enc_r[:, :, 0] != pad).float(), (dec_r[:, :, 0] != pad).float()

        pf = not os.environ.get('PB_NO_PACK_PREFETCH')
        if pf:
            eng.prefetch_pack(lm_r, em_r, dm_r)                      # as in the headline loop (and the Pretrainer): the next batch's row counts are requested a step ahead

        def real_step():
            if pf:
                eng.prefetch_pack(lm_r, em_r, dm_r)
            eng.loss_and_grads(enc_r, dec_r, tgt16, lm_r, em_r, dm_r, train=True, ids_checked=True)
            eng.optimizer_step(lr=2e-5, gscale=1.0)
        real_step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            real_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        Te, Td, _, Ts = eng.last_rows
        fl = train_flops_live_rows(Te, Td, eng.last_pairs, args.hs, args.layers, args.ffn, Ts=Ts)
        out["real_mix_step"] = {"ms_per_step": ms, "rows": {"encoder_side": Te, "decoder_side": Td, "last_layer_query_side": Ts, "of": B * S},
                                "loss_rows_fraction": Ts / float(Td), "step_mfma_frac": fl / (ms * 1e-3) / 1e12 / peak,
                                "tokens_per_s": B * S / (ms * 1e-3), "corruption_choices": [int(x) for x in ch.cpu()],
                                "note": "reference corruption mix (pb_corrupt, choices 1..5 per sample, seed 1234) of the same clean sequences; 10 steps"}
    except Exception as ex:                                            # noqa: BLE001
        out["real_mix_step"] = {"error": repr(ex)[:200]}
    # (a'') what the step EXCLUDES and SURVEY 8(d) wants reported beside it: building the batch on the device -- the reference's gen_mask over
    # the B sequences (pretrain.py:131-144: serial Python, 20-80 ms per SAMPLE), here ONE pb_corrupt launch + shift-right + the two masks.
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 20
        torch.cuda.synchronize()
        e0.record()
        for _ in range(nrep):
            ops.corrupt(tgt16, enc_r, lm_r, ch, None, 0.15, 0x1234ABCD, pb.pad_word_np, pb.mask_word_np, pb.n_tokens)
            ops.shift_right(tgt16, eng.sos16, dec_r, B, S)
            em_r, dm_r = (enc_r[:, :, 0] != pad).float(), (dec_r[:, :, 0] != pad).float()
        e1.record()
        torch.cuda.synchronize()
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e2.record()
        for _ in range(nrep):
            ops.corrupt(tgt16, enc_r, lm_r, ch, None, 0.15, 0x1234ABCD, pb.pad_word_np, pb.mask_word_np, pb.n_tokens)
        e3.record()
        torch.cuda.synchronize()
        out["corruption_ms"] = {"ms_per_batch": e0.elapsed_time(e1) / nrep, "pb_corrupt_kernel_ms": e2.elapsed_time(e3) / nrep, "batch": B, "seq_len": S,
                                "bytes_per_batch": B * S * 8 * (2 + 2 + 4), "launches": 4,
                                "note": "excluded from `value` (SURVEY 8d): pb_corrupt (reference 5-way mix, one workgroup per sequence) + pb_shift_right + "
                                        "2 mask compares for the B = %d batch, HIP events over %d repetitions; the reference does this as serial Python "
                                        "per sample on the host (pretrain.py:131-144)" % (B, nrep)}
    except Exception as ex:                                            # noqa: BLE001
        out["corruption_ms"] = {"error": repr(ex)[:200]}
    with stdout_to_stderr():                                         # the first communicator's banner goes to stderr
        try:
            from pianobart_amd.parallel import GradReducer
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
            own_pg = not dist.is_initialized()
            if own_pg:
                dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
            red = GradReducer(eng, 1)
            try:
                pfd = not os.environ.get('PB_NO_PACK_PREFETCH')
                if pfd:
                    eng.prefetch_pack(*step.batch[3:6])

                def dp_step():
                    if pfd:
                        eng.prefetch_pack(*step.batch[3:6])          # the headline loop's pipeline hint: without it every step drains the stream for its row counts
                    eng.loss_and_grads(*step.batch, train=True, count_hook=red.reduce_counts, ids_checked=True)
                    red.all_reduce_grads()
                    eng.optimizer_step(lr=2e-5, gscale=1.0)
                dp_step(); dp_step(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    dp_step()
                torch.cuda.synchronize()
                out["dp_mode_step"] = {"ms_per_step": (time.perf_counter() - t0) / 10 * 1e3, "world": 1, "exchange": red.mode,
                                       "note": "GradReducer installed at world size 1: per-layer bucket exchange through RCCL on the communication "
                                               "stream, backward GEMMs as ordinary grids; 10 steps"}
            finally:
                red.close() if hasattr(red, 'close') else None
                eng.grad_hook = None
                if own_pg:
                    dist.destroy_process_group()
        except Exception as ex:                                            # noqa: BLE001
            eng.grad_hook = None
            out["dp_mode_step"] = {"error": repr(ex)[:200]}
    try:
        biases = [p.detach().clone() for p in model.mask_lm.proj.parameters()]
        model.eval()
        with torch.no_grad():
            for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
                model.mask_lm.proj[i].bias[p0:] = -30.0                # special ids unsamplable: the loop runs for as long as we let it
        eng.refresh_shadow(force=True)
        enc = synth_octuple_batch(1, S, seed=7, min_len=S // 2)[5].to(dev)
        emask = (enc[:, :, 0] != 256).float()
        ntok = min(200, S)
        sampler = dict(T=model.SAMPLE_T, P=model.SAMPLE_P)           # model.sample_row IS model.py:68-107 with these constants: the device may sample ahead
        np.random.seed(0)
        eng.generate(enc[:, :64].contiguous(), emask[:, :64].contiguous(), lambda r: torch.tensor([256, 128, 129, 256, 128, 32, 254, 49]))
        eng.generate(enc, emask, model.sample_row, max_new=16, sampler=sampler)      # warm-up of the device-sampled path (graph capture)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(enc, emask, model.sample_row, max_new=ntok, sampler=sampler)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        info = eng.last_decode or {}
        d, f, L = args.hs, args.ffn, args.layers
        n = max(1, info.get('tokens', ntok + 1))
        ms_tok = info.get('loop_ms', wall) / n
        s_enc = info.get('s_enc', S // 2)
        wbytes = 2.0 * (L * (8 * d * d + 2 * d * f) + 1280 * d)         # bf16 decoder-side weights read once per token
        kvbytes = 2.0 * L * 2 * d * (s_enc + n / 2.0)                   # cross K/V rows + the self-attention rows decoded so far (mean)
        out["decode"] = {"ms_per_token": ms_tok, "tokens": n, "launches_per_token": info.get('launches_per_token'), "graph": info.get('graph'),
                         "prompt_ms_encoder_and_cross_kv": wall - info.get('loop_ms', wall), "bytes_per_token": wbytes + kvbytes,
                         "hbm_gbps": (wbytes + kvbytes) / (ms_tok * 1e-3) / 1e9, "hbm_frac": (wbytes + kvbytes) / (ms_tok * 1e-3) / 8e12,
                         "device_sampler": info.get('device_sampler', False), "rewinds": info.get('rewinds'),
                         "tokens_per_graph_replay": info.get('tokens_per_graph_replay', 1),
                         "note": "BASELINE configs[3] shape (B = 1, S = %d, %d visible encoder rows): per-token wall time of the decode loop. The "
                                 "device samples each position itself from uniform draws made ahead (8 tokens per hipGraph replay); the host "
                                 "replays every position from the logged logits row with the reference's sampler and RNG stream and rewinds the "
                                 "decoder where it disagrees (`rewinds`), so the tokens are the host loop's" % (S, s_enc)}
        with torch.no_grad():
            for p_, b_ in zip(model.mask_lm.proj.parameters(), biases):
                p_.copy_(b_)
        eng.refresh_shadow(force=True)
        model.train()
    except Exception as ex:                                            # noqa: BLE001
        out["decode"] = {"error": repr(ex)[:200]}
    return out


def side_model_step(cfgkw, precision, B, S, dev, nsteps, seed=1234):
    """One more model in the same process (its own parameters, engine and synthetic SURVEY 8(d) batch), `nsteps` timed steps after 2
    warm-up steps; returns (ms per step, executed FLOPs per step, kept rows). Used for the legs whose shape or precision differs from the
    headline's: the exact-f32 parity instantiation and BASELINE configs[4]'s model on one GPU. Freed before returning."""
    from pianobart_amd import ops
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab, synth_octuple_batch
    e2w, w2e = load_vocab()
    torch.manual_seed(1)
    model = PianoBartLM(PianoBart(BartConfig(**cfgkw), e2w, w2e, precision=precision)).train().to(dev)
    eng = model._get_engine()
    eng.bind(dev)
    eng.pipeline_updates = True
    enc, dec, loss_mask, emask, dmask, target = [x.to(dev) for x in synth_octuple_batch(B, S, seed)]
    b = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)

    eng.prefetch_pack(b[3], b[4], b[5])                           # the headline loop's pipeline hint (a no-op for instantiations that do not pack)

    def one():
        eng.prefetch_pack(b[3], b[4], b[5])
        eng.loss_and_grads(*b, train=True, ids_checked=True)
        eng.optimizer_step(lr=2e-5, gscale=1.0)
    one(); one(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / nsteps * 1e3
    Te, Td, _, Ts = eng.last_rows
    d, L, f = cfgkw['d_model'], cfgkw['encoder_layers'], cfgkw['encoder_ffn_dim']
    fl = train_flops_live_rows(Te, Td, eng.last_pairs, d, L, f, Ts=Ts)
    rows = {"encoder_side": Te, "decoder_side": Td, "last_layer_query_side": Ts, "of": B * S}
    eng.finish_updates()
    del eng, model, b
    torch.cuda.empty_cache()
    return ms, fl, rows


def other_instantiations(args, cfgkw, dev):
    """SURVEY 8(d) legs the headline does not cover, each driver-timed in the default run:
    fp32_parity_step -- the SAME model shape in the exact-f32 instantiation (`precision="fp32"`: f32 storage, f32-input MFMA = an fmaf chain,
    unfused attention), i.e. the instantiation that is gated at the north-star tolerance (logits 1e-4 against the CPU oracle), at B = 4;
    cfg5_step -- BASELINE configs[4]'s model (24L / 1024d / ffn 4096 / 16 heads, S = 2048) on ONE GPU at B = 8, bf16."""
    out = {}
    try:
        B4 = 4
        ms, fl, rows = side_model_step(cfgkw, 'fp32', B4, args.seq, dev, 3)
        out["fp32_parity_step"] = {"ms_per_step": ms, "batch": B4, "tokens_per_s": B4 * args.seq / (ms * 1e-3), "rows": rows,
                                   "step_tflops": fl / (ms * 1e-3) / 1e12, "peak_tflops_f32_mfma": 157.3, "frac_of_f32_mfma_peak": fl / (ms * 1e-3) / 1e12 / 157.3,
                                   "note": "precision='fp32' (exact f32 arithmetic, the parity-gated instantiation: tests/test_model_gpu.py at 1e-4), same "
                                           "model shape as the headline, B = 4, padded rows (the packed step is bf16-only), 3 steps"}
    except Exception as ex:                                            # noqa: BLE001
        out["fp32_parity_step"] = {"error": repr(ex)[:200]}
    try:
        for B3 in (4, 16, args.batch):
            ms, fl, rows = side_model_step(cfgkw, 'bf16x3', B3, args.seq, dev, 3)
            key = "bf16x3_parity_step" if B3 == 4 else "bf16x3_parity_step_b%d" % B3
            out[key] = {"ms_per_step": ms, "batch": B3, "tokens_per_s": B3 * args.seq / (ms * 1e-3), "rows": rows,
                        "step_tflops_algorithmic": fl / (ms * 1e-3) / 1e12, "frac_of_bf16_mfma_peak_algorithmic": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                        "note": "precision='bf16x3': f32 storage and row kernels, every GEMM as split-bf16 triples (a_hi b_hi + a_hi b_lo + a_lo b_hi, f32 "
                                "accumulate) on the bf16 MFMA -- 3x the matrix work per algorithmic FLOP; logits within 1e-3 of the CPU reference "
                                "(tests/test_model_gpu.py: G1, G4, G10: measured 2e-5); same model shape, dead rows dropped like the bf16 step, fused split-bf16 attention, 3 steps"}
        if "ms_per_step" in out.get("fp32_parity_step", {}):
            out["bf16x3_parity_step"]["speedup_over_fp32_parity_step"] = out["fp32_parity_step"]["ms_per_step"] / out["bf16x3_parity_step"]["ms_per_step"]
        ms32, _, _ = side_model_step(cfgkw, 'fp32', 16, args.seq, dev, 2)            # the exact-f32 step at the larger batch, for the same ratio there
        out["bf16x3_parity_step_b16"]["fp32_ms_per_step_same_batch"] = ms32
        out["bf16x3_parity_step_b16"]["speedup_over_fp32_same_batch"] = ms32 / out["bf16x3_parity_step_b16"]["ms_per_step"]
    except Exception as ex:                                            # noqa: BLE001
        out.setdefault("bf16x3_parity_step", {"error": repr(ex)[:200]})
    if args.precision == 'bf16' and (args.layers, args.hs, args.seq) == (12, 768, 1024):
        try:
            kw5 = dict(max_position_embeddings=2048, d_model=1024, encoder_layers=24, decoder_layers=24, encoder_ffn_dim=4096, decoder_ffn_dim=4096,
                       encoder_attention_heads=16, decoder_attention_heads=16, dropout=cfgkw['dropout'])
            B5, S5 = 8, 2048
            ms, fl, rows = side_model_step(kw5, 'bf16', B5, S5, dev, 5)
            out["cfg5_step"] = {"ms_per_step": ms, "batch": B5, "seq_len": S5, "tokens_per_s": B5 * S5 / (ms * 1e-3), "rows": rows,
                                "step_tflops": fl / (ms * 1e-3) / 1e12, "step_mfma_frac": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                "flops_per_token_train": train_flops_per_token(S5, 1024, 24, 4096),
                                "note": "BASELINE configs[4]'s model (24L / 1024d / ffn 4096 / 16 heads, S = 2048) on one GPU, B = 8, bf16, dropout on, packed rows, 5 steps"}
        except Exception as ex:                                        # noqa: BLE001
            out["cfg5_step"] = {"error": repr(ex)[:200]}
    return out


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints its version banner on file descriptor 1 when the first communicator comes up; stdout carries the ONE JSON line and nothing
    else, so descriptor 1 points at stderr while a communicator is being created."""
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)                             # the banner sits in libc's stdout buffer (a pipe / file is fully buffered)
        except Exception:                                              # noqa: BLE001
            pass
        os.dup2(fd_out, 1)
        os.close(fd_out)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _kfd_gpu_count():
    """GPUs of this node without a HIP call: KFD topology nodes with SIMDs (CPU nodes have simd_count 0), cut down to the
    ROCR_ / HIP_VISIBLE_DEVICES list when one is set; torch's count if the topology cannot be read."""
    try:
        base, n = '/sys/class/kfd/kfd/topology/nodes', 0
        for node in os.listdir(base):
            props = dict(l.split(None, 1) for l in open(os.path.join(base, node, 'properties')).read().splitlines() if ' ' in l)
            n += int(props.get('simd_count', '0')) > 0
        for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
            if os.environ.get(var, '').strip():
                n = min(n, len([v for v in os.environ[var].split(',') if v.strip()]))
        return n
    except (OSError, ValueError):
        return torch.cuda.device_count()


def launch_ranks(n, argv, device_count=None, run=None, out=None):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: this process becomes a LAUNCHER. It never imports
    pianobart_amd; it counts the devices from /sys/class/kfd (no HIP call -- torch.cuda.device_count() may bring the HIP runtime up when
    amdsmi is unavailable, ADVICE r5, so it is only the fallback) and in any case STARTS A CHILD and never replaces itself: do not turn
    the child start into an os.exec* on the strength of "the launcher has not touched the GPU". It starts
        python -m torch.distributed.run --nnodes 1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>
    as a CHILD process (one rank per GPU over RCCL, the reference's nn.DataParallel replaced: pretrain.py:63-65), relays rank 0's ONE
    JSON line to stdout and returns the child's exit code -- the line is NOT relayed when the child failed (a consumer that reads stdout
    without looking at the exit code must not see a line from a failed job). A box with fewer than N GPUs is refused loudly: the line
    must never say n_gpus 1 for a --gpus N request. `device_count`, `run` and `out` are injection points for the CPU test of this logic."""
    import subprocess
    have = (_kfd_gpu_count() if device_count is None else device_count)
    if have < n:
        sys.stderr.write('bench.py: --gpus %d requested but this node shows %d GPU(s): refusing to run (no silent fallback to fewer ranks)\n' % (n, have))
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes', '1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = (run or subprocess.run)(cmd, stdout=subprocess.PIPE, text=True, env=env)          # stderr passes through
    lines = [l for l in (r.stdout or '').splitlines() if l.startswith('{')]
    out = out or sys.stdout
    if lines:
        try:
            rec = json.loads(lines[-1])
        except ValueError:
            rec = None
        if rec is not None and r.returncode == 0 and rec.get('n_gpus') != n:
            sys.stderr.write('bench.py: the ranks reported n_gpus %r for --gpus %d\n' % (rec.get('n_gpus'), n))
            return 3
        if r.returncode != 0:
            sys.stderr.write('bench.py: the ranks exited %d: their JSON line is withheld\n' % r.returncode)
            return r.returncode
        out.write(lines[-1] + '\n'); out.flush()
    elif r.returncode == 0:
        sys.stderr.write('bench.py: the %d ranks exited 0 without a JSON line\n' % n)
        return 4
    return r.returncode


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == '--cpu-baseline-child':
        return cpu_baseline_child(json.loads(sys.argv[2]), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (weak scaling)')
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--layers', type=int, default=12)
    ap.add_argument('--hs', type=int, default=768)
    ap.add_argument('--ffn', type=int, default=3072)
    ap.add_argument('--heads', type=int, default=12)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'bf16x3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dropout', action='store_true')
    ap.add_argument('--no-probe', action='store_true', help='skip the per-family in-step timing that follows the timed region')
    ap.add_argument('--force-reducer', action='store_true', help='install the RCCL gradient reducer even at world size 1 (test)')
    ap.add_argument('--mode', default='pretrain', choices=['pretrain', 'decode'], help='decode = BASELINE configs[3]: KV-cached generate, B=1')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:               # plain `python bench.py --gpus N`: start the N ranks as a child job
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0)); local_rank = int(os.environ.get('LOCAL_RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: the launcher and the flag disagree (refusing to report a line for another N)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path): torch.cuda.is_available() is False')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or args.force_reducer:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
        with stdout_to_stderr():
            dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)
            warm = torch.ones(1, device=dev)
            dist.all_reduce(warm)                                    # the communicator (and its banner) comes up here at the latest
            torch.cuda.synchronize()

    from pianobart_amd import ops
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from pianobart_amd.parallel import GradReducer
    from tests.golden_util import load_vocab
    e2w, w2e = load_vocab()
    cfgkw = dict(max_position_embeddings=args.seq, d_model=args.hs, encoder_layers=args.layers, decoder_layers=args.layers,
                 encoder_ffn_dim=args.ffn, decoder_ffn_dim=args.ffn, encoder_attention_heads=args.heads,
                 decoder_attention_heads=args.heads, dropout=0.0 if args.no_dropout else 0.1)
    torch.manual_seed(0)
    model = PianoBartLM(PianoBart(BartConfig(**cfgkw), e2w, w2e, precision=args.precision)).train().to(dev)
    eng = model._get_engine()
    eng.bind(dev)
    if args.mode == 'decode':
        return decode_bench(args, model, eng, dev, rank)
    B, S = args.batch, args.seq
    enc, dec, loss_mask, emask, dmask, target = synth_rank_batch(B, S, world, rank, dev)
    enc16, dec16, tgt16 = ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target)
    loss_mask = loss_mask.contiguous()
    reducer = GradReducer(eng, world) if (world > 1 or args.force_reducer) else None
    eng.pipeline_updates = not os.environ.get('PB_NO_PIPELINE_UPDATES')     # the parameter update runs beside the next step's forward (Engine.optimizer_step)

    prefetch = not os.environ.get('PB_NO_PACK_PREFETCH')
    if prefetch:
        eng.prefetch_pack(loss_mask, emask, dmask)                  # the first step's own request

    def step():
        # the data pipeline knows the next batch while this one is being enqueued (here: the same resident tensors): its row counts are
        # requested now, so the next step's packing plan does not wait for this step to drain (Engine.prefetch_pack)
        if prefetch:
            eng.prefetch_pack(loss_mask, emask, dmask)              # for the step AFTER this one (this one's request went out a step ago)
        sums = eng.loss_and_grads(enc16, dec16, tgt16, loss_mask, emask, dmask, train=True, ids_checked=True,      # ids generated in range (synth_batch)
                                  count_hook=reducer.reduce_counts if reducer else None)
        if reducer:
            reducer.all_reduce_grads()
        eng.optimizer_step(lr=2e-5, gscale=1.0)
        return sums
    step.batch = (enc16, dec16, tgt16, loss_mask, emask, dmask)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        sums = step()
        marks[i + 1].record()                       # HIP event on the launch stream after the optimizer (the second stream has joined by then)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if os.environ.get('PB_LONG_RUN_DUMP') and rank == 0:             # developer aid (tools/long_run.py): the step times in order
        with open(os.environ['PB_LONG_RUN_DUMP'], 'w') as fh:
            json.dump(per_step, fh)
    per_step.sort()
    ms_median = per_step[len(per_step) // 2]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    rccl_ranks = 1
    if world > 1 or args.force_reducer:                              # what the collective library itself says about the job's size
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
    ms_per_step = dt / args.steps * 1e3
    tokens = B * S * world * args.steps
    value = tokens / dt
    fpt = train_flops_per_token(S, args.hs, args.layers, args.ffn)
    ref_tflops_per_gpu = fpt * B * S / (ms_per_step * 1e-3) / 1e12               # the reference graph: every padded row credited
    peak = PEAK_BF16_TFLOPS if args.precision in ('bf16', 'bf16x3') else 157.3         # bf16x3 runs on the bf16 matrix cores (3 MFMAs per algorithmic product)
    T = B * S
    Te, Td, _, Ts = eng.last_rows
    live_flops = train_flops_live_rows(Te, Td, eng.last_pairs, args.hs, args.layers, args.ffn, Ts=Ts)
    step_tflops_per_gpu = live_flops / (ms_per_step * 1e-3) / 1e12              # the rows and (query, key) pairs the step computes

    # ---- roofline of the dominant kernel AS LAUNCHED BY THE STEP (fc1: NT T x ffn x d + bias + GELU + derivative out), and the
    # per-family shares, from HIP events on the launch stream (one-stream schedule; after the timed region, rank 0 at N=1 only)
    families = table = None
    fc1_ms = fc1_n = None
    if world == 1 and not args.no_probe:
        from pianobart_amd import engine as E
        saved = E._WGRAD_STREAM
        E._WGRAD_STREAM = 0
        try:
            step(); torch.cuda.synchronize()
            nprobe = 3
            with StepProbe(ops, eng.last_pairs) as probe:
                probe.mark()
                for _ in range(nprobe):
                    step()
                torch.cuda.synchronize()
                families, table, shapes = probe.summary(nprobe, peak)
            if os.environ.get('PB_PROBE_DUMP'):                      # developer aid: every op label of the probed steps, not only the top 12
                with open(os.environ['PB_PROBE_DUMP'], 'w') as fh:
                    for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
                        fh.write('%9.1f us x %5.1f  %8.1f TF  %s\n' % (1e3 * v[0] / v[2], v[2] / nprobe, (v[1] / (v[0] * 1e-3) / 1e12) if v[1] else 0.0, k))
            key = [k for k in shapes if any(k.startswith('gemm NT %dx%dx%d+bias+gelu' % (m_, args.ffn, args.hs)) for m_ in {Te, Td})]
            if key:                                                   # fc1 of the encoder (M = Te) and decoder (M = Td) layers
                n_ = sum(shapes[k][2] for k in key)
                fc1_ms, fc1_n = sum(shapes[k][0] for k in key) / n_, n_ / nprobe
                fc1_flops = sum(shapes[k][1] for k in key) / n_
        finally:
            E._WGRAD_STREAM = saved
    # the same kernel alone, back to back on random operands with its real epilogue, AT THE ROW COUNT THE STEP LAUNCHES IT WITH (the encoder
    # side's Te; until round 4 this probe ran the padded T = B S rows, which is most of the "0.244 alone vs 0.169 in the step" of that round's
    # record: 32768 / 26624 rows). At equal rows the loop still runs ~20 % slower per launch than the step's launches (round 5: 0.203 against
    # 0.168 ms on one box, host enqueue 0.009 ms per launch, so not the host): 20 launches in a row each write 2 x Te x ffn x 2 B with no lighter
    # neighbour kernel in between.
    code = ops.dtype_code(eng.xdt)
    iso_ms, iso_host_ms, Tiso = None, None, (Te if fc1_ms else T)
    if not args.no_probe:
        iso_ms, iso_host_ms = isolated_fc1(ops, eng, args, code, Tiso, dev)
    gemm_ms = fc1_ms if fc1_ms else iso_ms
    gemm_tflops = ((fc1_flops if fc1_ms else 2.0 * Tiso * args.ffn * args.hs) / (gemm_ms * 1e-3) / 1e12) if gemm_ms else None

    # ---- numbers the driver's default line would otherwise never see: each a few seconds, all after the timed region
    extras = {}
    if world == 1 and not args.no_probe:
        extras = extra_measurements(args, model, eng, ops, step, peak, dev)
        extras.update(other_instantiations(args, cfgkw, dev))
    # the largest single row of the committed kernel trace is not the GEMM but the one-pass attention backward: a second roofline entry
    # for it, from the same in-step HIP events (the interval of a flash_bwd1_packed call = fa1_bwd + its fa1_reduce launch)
    roofline2 = None
    if table is not None:
        fa = [(k, v) for k, v in shapes.items() if k.startswith(('flash_bwd1_packed', 'flash_bwd1 '))]
        if fa:
            ms_ = sum(v[0] for _, v in fa); fl_ = sum(v[1] for _, v in fa); n_ = sum(v[2] for _, v in fa)
            roofline2 = {"bound": "mfma", "achieved": fl_ / (ms_ * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl_ / (ms_ * 1e-3) / 1e12 / peak,
                         "traffic": None, "kernel": "fa1_bwd_kernel (one-pass attention backward, head_dim 64) + fa1_reduce, encoder self- and cross-attention calls",
                         "avg_launch_ms": ms_ / n_, "calls_per_step": n_ / nprobe,
                         "how": "HIP events around the %d calls of the probe steps; FLOPs = 2.5 x 4 x H x hd x (query, key) pairs covered" % n_}

    if rank == 0:
        traffic, traffic_src = pmc_traffic(sorted({Te, Td}), args.ffn, args.hs)
        s = sums.double().cpu()
        w8 = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
        loss = float(((s[0:8] / s[8:16]) * w8).sum() / w8.sum())
        rec = {
            "metric": "Octuple tokens/sec/GPU (seq=1024, 12L/768d) pretrain step; %MFMA peak",
            "value": value, "unit": "tokens/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median_hip_events": ms_median, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "tokens_per_s_per_gpu": value / world,
            "step_tflops_per_gpu": step_tflops_per_gpu, "step_mfma_frac": step_tflops_per_gpu / peak,
            "reference_graph_tflops_per_gpu": ref_tflops_per_gpu,
            "rows": {"encoder_side": Te, "decoder_side": Td, "last_decoder_layer_query_side_and_heads": Ts, "padded": T,
                     "note": "dead-row compaction: rows that are neither visible as attention keys nor carry a loss term are dropped from the "
                             "step, and the last decoder layer's cross-attention block, FFN and the LM heads run on the rows with a loss term "
                             "only (results unchanged, tests/test_packed_gpu.py); `value` counts all B x S tokens of the batch, the unit the "
                             "reference's own tokens/s is in; step_tflops_per_gpu / step_mfma_frac / roofline count only the rows and "
                             "(query, key) pairs that are computed; PB_PACK_ROWS=0 runs the dense step"},
            "train_loss": loss,
            "config": {"workload": "pretrain step %dL/%dd/ffn%d/%dh S=%d B=%d/GPU dropout=%s (BASELINE configs[1])" %
                       (args.layers, args.hs, args.ffn, args.heads, S, B, cfgkw['dropout']),
                       "global_batch": B * world, "seq_len": S, "parallelism": "dp%d" % world,
                       "flops_per_token_train": fpt,
                       "flops_note": "flops_per_token_train: algorithmic FLOPs of the reference graph (SURVEY 8d), full S^2 attention (causal at "
                                     "1/2), every padded row; step_tflops_per_gpu uses the same graph restricted to the rows / pairs the step computes"},
            "roofline": {"bound": "mfma", "achieved": gemm_tflops, "peak": peak, "unit": "TFLOP/s", "frac": (gemm_tflops / peak) if gemm_tflops else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "gemm3_kernel<%s,NT> (256x256 ping-pong) fc1 M=%s N=%d K=%d + bias + GELU + derivative out" %
                                   (args.precision, T if Te == T else '%d (encoder layers) / %d (decoder layers)' % (Te, Td), args.ffn, args.hs),
                         "avg_launch_ms": gemm_ms,
                         "how": ("HIP events around each of the %.0f fc1 launches of a step, inside the step (one-stream schedule)" % fc1_n) if fc1_ms
                                else "isolated back-to-back launches (probe disabled)",
                         "isolated_back_to_back_ms": iso_ms, "isolated_rows": Tiso, "isolated_host_enqueue_ms_per_launch": iso_host_ms,
                         "families_in_step": families, "top_ops_in_step": table},
        }
        if roofline2 is not None:
            rec["roofline_secondary"] = roofline2
        rec.update(extras)
        rec["rows"]["loss_rows_fraction"] = Ts / float(Td)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfgkw, S)
        print(json.dumps(rec), flush=True)
    if world > 1 or args.force_reducer:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
