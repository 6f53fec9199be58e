"""Where a decode token's wall time goes: host time inside pb_decode_step (98 launches), GPU time between its first and last
kernel (HIP events), and the rest (D2H logits, host nucleus sampling, H2D token)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pianobart_amd import _lib
from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
from tests.golden_util import load_vocab, synth_octuple_batch
e2w, w2e = load_vocab()
kw = dict(max_position_embeddings=1024, d_model=768, encoder_layers=12, decoder_layers=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
          encoder_attention_heads=12, decoder_attention_heads=12)
m = PianoBartLM(PianoBart(BartConfig(**kw), e2w, w2e)).eval().cuda()
with torch.no_grad():
    for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
        m.mask_lm.proj[i].bias[p0:] = -30.0
eng = m._get_engine()
enc = synth_octuple_batch(1, 1024, seed=7, min_len=512)[5].cuda(); em = (enc[:, :, 0] != 256).float()
N = 400
cnt = {'n': 0}
def sampler(row):
    cnt['n'] += 1
    return m.sample_row(row) if cnt['n'] <= N else torch.tensor([256, 128, 129, 256, 128, 32, 254, 49])
host, evs = [], []
orig = _lib.LIB.call
def call(name, *a):
    if name != 'pb_decode_step':
        return orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t0 = time.perf_counter(); orig(name, *a); host.append(time.perf_counter() - t0); e1.record(); evs.append((e0, e1))
_lib.LIB.call = call
np.random.seed(0)
eng.generate(enc[:, :64].contiguous(), em[:, :64].contiguous(), lambda r: torch.tensor([256, 128, 129, 256, 128, 32, 254, 49]))
host.clear(); evs.clear()
torch.cuda.synchronize(); t0 = time.perf_counter(); eng.generate(enc, em, sampler); torch.cuda.synchronize(); wall = time.perf_counter() - t0
gpu = [a.elapsed_time(b) * 1e3 for a, b in evs[5:N]]
print('per token over %d tokens: wall %.1f us | host inside pb_decode_step %.1f us | GPU first->last kernel %.1f us (median %.1f) | rest (D2H, sampling, H2D) %.1f us'
      % (N, wall / (N + 1) * 1e6, np.mean(host[5:N]) * 1e6, np.mean(gpu), np.median(gpu), wall / (N + 1) * 1e6 - np.mean(gpu)))
