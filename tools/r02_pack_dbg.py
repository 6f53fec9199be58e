"""Developer aid: packed vs dense fused step on a small model, printing where they differ."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from pianobart_amd import engine as E, ops
from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
from tests.golden_util import load_vocab, randomize_params, synth_octuple_batch

e2w, w2e = load_vocab()
B, S, d, heads = 6, 256, 256, 4
cfg = BartConfig(max_position_embeddings=S, d_model=d, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                 encoder_attention_heads=heads, decoder_attention_heads=heads, dropout=0.0)
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
    dmask[b, :Ld[b]] = 1; dmask[b, Ld[b]:] = 0
    loss_mask[b, Ld[b]:] = 0
    loss_mask[b, :Ld[b], 0] = 1
args = (ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(target), loss_mask.contiguous(), emask, dmask)
res = []
for pack in (0, 1):
    E._PACK_ROWS = pack
    eng._seed = 77
    s = eng.loss_and_grads(*args, train=True).clone()
    torch.cuda.synchronize()
    ws = eng._cur_ws
    pk = eng._saved.get('pack')
    # final decoder hidden and encoder output, scattered back to (B*S) rows for comparison
    dec_out, enc_out = eng._saved['dec_out'].float().clone(), eng._saved['enc_out'].float().clone()
    if pk is not None:
        full_d = torch.zeros(B * S, d, device='cuda'); full_e = torch.zeros(B * S, d, device='cuda')
        src_d = eng._pack_state['src_d'][:pk.Td].long(); src_e = eng._pack_state['src_e'][:pk.Te].long()
        full_d[src_d] = dec_out; full_e[src_e] = enc_out
        keep_d = torch.zeros(B * S, dtype=torch.bool, device='cuda'); keep_d[src_d] = True
        keep_e = torch.zeros(B * S, dtype=torch.bool, device='cuda'); keep_e[src_e] = True
        dec_out, enc_out = full_d, full_e
    else:
        keep_d = keep_e = None
    res.append((s, eng.G32.clone(), dec_out, enc_out, keep_d, keep_e, eng.last_rows))
(s0, g0, d0, e0, _, _, r0), (s1, g1, d1, e1, kd, ke, r1) = res
print('rows', r0, r1)
print('sums dense ', s0.tolist())
print('sums packed', s1.tolist())
vis_e = (emask.reshape(-1) != 0)
print('enc_out diff on visible rows', float((e0 - e1)[vis_e & ke].abs().max()), 'all visible kept', bool((ke | ~vis_e).all()))
live_d = (dmask.reshape(-1) != 0)
print('dec_out diff on visible rows', float((d0 - d1)[live_d & kd].abs().max()), 'all kept', bool((kd | ~live_d).all()))
for name, sl in eng.slots.items():
    a, b_ = g0[sl.off:sl.off + sl.numel], g1[sl.off:sl.off + sl.numel]
    err = float((a - b_).norm()) / (float(a.norm()) + 1e-12)
    if err > 1e-4:
        print('%-16s %.3e' % (name, err))
