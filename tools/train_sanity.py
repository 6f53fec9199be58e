"""Sanity: bf16 fused step on one fixed cfg-2 batch (B=8) for 40 steps at lr 3e-4: the loss must fall steadily.
Also times Pretrainer.prepare_batch (device corruption + shift + masks) at B=32, S=1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
from pianobart_amd.pretrain import Pretrainer
from tests.golden_util import load_vocab, synth_octuple_batch
e2w, w2e = load_vocab()
kw = dict(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
          encoder_attention_heads=12, decoder_attention_heads=12, dropout=0.1)
torch.manual_seed(0)
tr = Pretrainer(PianoBart(BartConfig(**kw), e2w, w2e, precision='bf16'), None, None, 3e-4, 8, 1024, 0.15, False, [0])
tr.model.train()
batch = synth_octuple_batch(32, 1024, seed=5)[5]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    prep = tr.prepare_batch(batch.cuda())
torch.cuda.synchronize(); print('prepare_batch B=32 S=1024: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
enc16, dec16, tgt16, lm, em, dm = [t[:8].contiguous() for t in prep]
w = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
for it in range(40):
    s = tr.engine.loss_and_grads(enc16, dec16, tgt16, lm, em, dm, train=True)
    tr.engine.optimizer_step(lr=3e-4)
    if it % 5 == 0 or it == 39:
        s = s.double().cpu(); print('step %2d loss %.4f acc %.3f' % (it, float(((s[0:8] / s[8:16]) * w).sum() / w.sum()), float((s[16:24] / s[8:16]).mean())), flush=True)
