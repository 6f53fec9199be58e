"""End-to-end throughput of Pretrainer.iteration (data loader -> H2D -> device-side corruption -> fused step -> log line) at the
bench shape on synthetic Octuple shards, next to what bench.py reports for the step alone.
  python tools/pretrainer_e2e.py [--workers N] [--batches K]"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pianobart_amd import model as M
from pianobart_amd.data import MidiDataset
from pianobart_amd.pretrain import Pretrainer
from tests.golden_util import load_vocab, synth_octuple_batch

ap = argparse.ArgumentParser()
ap.add_argument('--workers', type=int, default=5)
ap.add_argument('--batches', type=int, default=24)
ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
e2w, w2e = load_vocab()
S, B = 1024, args.batch
X = synth_octuple_batch(B * args.batches, S, seed=5)[5].numpy().astype(np.int16)        # clean targets, PAD tails as in the bench batch
from torch.utils.data import DataLoader
from pianobart_amd.pretrain import _loader_kw
mk = lambda: DataLoader(MidiDataset(X=X), batch_size=B, shuffle=True, **_loader_kw(args.workers))
cfg = M.BartConfig(max_position_embeddings=S, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
                   encoder_attention_heads=12, decoder_attention_heads=12)
pb = M.PianoBart(cfg, e2w, w2e, precision='bf16')
tr = Pretrainer(pb, mk(), mk(), 2e-5, B, S, 0.15, False, [0])
tr.quiet = True
tr.iteration(tr.train_data, S, train=True)                       # warm-up epoch (allocations, second-stream probe)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.iteration(tr.train_data, S, train=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('Pretrainer.iteration: %d batches of %d x %d in %.2f s = %.2f ms/batch = %.0f tokens/s (workers=%d)' %
      (args.batches, B, S, dt, 1e3 * dt / args.batches, args.batches * B * S / dt, args.workers))

# the same loop fed from an in-memory list of batches (no DataLoader): what the loader costs
batches = [b for b in tr.train_data]
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.iteration(batches, S, train=True)
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print('  same batches from a list: %.2f ms/batch' % (1e3 * dt2 / len(batches)))
# the step alone on one of them, as bench.py runs it (resident, already corrupted inputs)
prep = tr.prepare_batch(batches[0])
eng = tr.engine
for _ in range(3):
    eng.loss_and_grads(*prep, train=True); eng.optimizer_step(lr=2e-5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.loss_and_grads(*prep, train=True); eng.optimizer_step(lr=2e-5)
torch.cuda.synchronize()
print('  step alone on one resident batch: %.2f ms (rows kept %s)' % (1e3 * (time.perf_counter() - t0) / 20, eng.last_rows))
