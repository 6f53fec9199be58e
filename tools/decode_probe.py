"""Where does a decode token go? native step GPU time vs host sampling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
fixed = torch.tensor([1, 2, 3, 4, 5, 6, 7, 8])
n = {'c': 0}
def s_fixed(row):
    n['c'] += 1
    return fixed if n['c'] <= 256 else torch.tensor([256, 128, 129, 256, 128, 32, 254, 49])
eng.generate(enc[:, :64].contiguous(), em[:, :64].contiguous(), lambda r: torch.tensor([256, 128, 129, 256, 128, 32, 254, 49]))
torch.cuda.synchronize(); t0 = time.perf_counter(); eng.generate(enc, em, s_fixed); torch.cuda.synchronize()
print('256 tokens, trivial sampler: %.3f ms/token (incl. encoder %.1f ms share)' % ((time.perf_counter() - t0) / 256 * 1e3, 0))
row = torch.randn(1280)
t0 = time.perf_counter()
for _ in range(50): m.sample_row(row)
print('host nucleus sampling (8 heads): %.3f ms/token' % ((time.perf_counter() - t0) / 50 * 1e3))
