"""Device-sampled decode over the WHOLE window (the last graph replays are shorter than 8 tokens, the position reaches S - 1): a 2-layer / 256-d model,
S = 1024, specials unsamplable so that nothing stops early; tokens and np.random state against the per-token host loop (PB_DECODE_SPEC=0).
  gpurun -- 'python tools/decode_full_window_check.py'   (round 6: equal, 1 024 tokens, 0 rewinds; loop 138.3 ms per-token path, 72.0 ms device-sampled)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pianobart_amd import engine as E
from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch
e2w, w2e = load_vocab()
S = 1024
kw = dict(max_position_embeddings=S, d_model=256, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512, encoder_attention_heads=4, decoder_attention_heads=4)
m = PianoBartLM(PianoBart(BartConfig(**kw), e2w, w2e, precision='bf16')).eval()
randomize_params(m, 31)
with torch.no_grad():
    for i, p0 in enumerate([256, 128, 129, 256, 128, 32, 254, 49]):
        m.mask_lm.proj[i].bias[p0:] = -30.0
m = m.cuda()
enc = synth_octuple_batch(1, S, seed=8, min_len=S - 9)[5].cuda()
emask = (enc[:, :, 0] != 256).float()
eng = m._get_engine()
res = {}
for spec in (0, 1):
    E._DECODE_SPEC = spec
    np.random.seed(3)
    out = eng.generate(enc, emask, m.sample_row, sampler=dict(T=m.SAMPLE_T, P=m.SAMPLE_P))
    res[spec] = (out.cpu(), np.random.get_state()[1].copy(), dict(eng.last_decode))
    print(spec, eng.last_decode)
print('equal tokens', torch.equal(res[0][0], res[1][0]), 'equal rng', np.array_equal(res[0][1], res[1][1]), 'tokens', res[1][2]['tokens'])
assert torch.equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[1][2]['tokens'] == S
