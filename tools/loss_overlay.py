"""bf16 against exact-f32 loss trajectories on the same data (SURVEY 7 hard part 1): the cfg-2 model (12L / 768d / ffn 3072 /
12 heads, S = 1024), B = 8, a fixed cycle of synthetic SURVEY 8(d) batches, identical initial weights, identical Philox dropout seeds, the fused
step + HF-AdamW of the pre-training loop (pretrain.py:179-196) for N optimizer steps in each precision.

  python tools/loss_overlay.py [--steps 200] [--batch 8] [--lr 1e-4] [--layers 12] [--out profiles/r05_loss_overlay.txt]

The throughput number of bench.py is the bf16 instantiation; the parity claims (logits <= 1e-3, exact argmax) are the exact-f32 one. This is the
curve that says what the bf16 run does to TRAINING: per step the two losses and their relative gap."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(precision, args, batches):
    from pianobart_amd import ops
    from pianobart_amd.model import BartConfig, PianoBart, PianoBartLM
    from tests.golden_util import load_vocab
    e2w, w2e = load_vocab()
    kw = dict(max_position_embeddings=args.seq, d_model=args.hs, encoder_layers=args.layers, decoder_layers=args.layers, encoder_ffn_dim=args.ffn,
              decoder_ffn_dim=args.ffn, encoder_attention_heads=args.heads, decoder_attention_heads=args.heads, dropout=args.dropout)
    torch.manual_seed(0)                                                   # the same initial weights in both runs
    m = PianoBartLM(PianoBart(BartConfig(**kw), e2w, w2e, precision=precision)).train().cuda()
    eng = m._get_engine()
    eng.bind(torch.device('cuda', 0))
    w8 = torch.tensor([262, 134, 262, 134, 38, 135, 55, 260], dtype=torch.double)
    dev = [[t.cuda() for t in b] for b in batches]
    prep = [(ops.ids_to_i16(enc), ops.ids_to_i16(dec), ops.ids_to_i16(tgt), lm.contiguous(), em, dm) for enc, dec, lm, em, dm, tgt in dev]
    losses = []
    for it in range(args.steps):
        s = eng.loss_and_grads(*prep[it % len(prep)], train=True, ids_checked=True)
        eng.optimizer_step(lr=args.lr)
        s = s.double().cpu()
        losses.append(float(((s[0:8] / s[8:16]) * w8).sum() / w8.sum()))
    torch.cuda.synchronize()
    del m, eng
    torch.cuda.empty_cache()
    return losses


def overlay(args):
    from tests.golden_util import synth_octuple_batch
    batches = [synth_octuple_batch(args.batch, args.seq, seed=100 + i) for i in range(args.nbatch)]
    lb = run(getattr(args, 'precision', 'bf16'), args, batches)
    lf = run('fp32', args, batches)
    return lb, lf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--nbatch', type=int, default=16, help='distinct batches, visited in a fixed cycle')
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--layers', type=int, default=12)
    ap.add_argument('--hs', type=int, default=768)
    ap.add_argument('--ffn', type=int, default=3072)
    ap.add_argument('--heads', type=int, default=12)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--out', default=None)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'bf16x3'], help='the instantiation laid over the exact-f32 one')
    args = ap.parse_args()
    lb, lf = overlay(args)
    gap = [abs(a - b) / b for a, b in zip(lb, lf)]
    tail = max(1, min(10, args.steps // 4))
    mean = lambda v: sum(v) / len(v)
    what = {'bf16': 'the throughput instantiation (bf16 storage / MFMA, f32 accumulation, f32 masters and moments)',
            'bf16x3': 'the split-bf16 parity instantiation (f32 storage, GEMMs and attention as bf16 triples on the bf16 MFMA)'}[args.precision]
    lines = ['# tools/loss_overlay.py --precision %s --steps %d --batch %d --nbatch %d --lr %g --layers %d --dropout %g   (12L/768d/ffn3072/12h unless said, S = %d; same initial '
             'weights, same batches in the same order, same Philox dropout seeds; fused step + HF-AdamW; 1x MI355X)' %
             (args.precision, args.steps, args.batch, args.nbatch, args.lr, args.layers, args.dropout, args.seq),
             '# loss_bf16 column = %s; loss_f32 = the exact-f32 parity instantiation' % what,
             '# max |gap| over the run %.3e (step %d); mean gap of the last %d steps %.3e; final losses %.6f (bf16) %.6f (f32); first %.6f / %.6f' %
             (max(gap), gap.index(max(gap)), tail, mean(gap[-tail:]), lb[-1], lf[-1], lb[0], lf[0]),
             '# step  loss_bf16  loss_f32  rel_gap']
    for i, (a, b, g_) in enumerate(zip(lb, lf, gap)):
        lines.append('%5d  %.6f  %.6f  %.2e' % (i, a, b, g_))
    text = '\n'.join(lines) + '\n'
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(text)
    print('\n'.join(lines[:4]))
    for i in list(range(0, args.steps, max(1, args.steps // 20))) + [args.steps - 1]:
        print(lines[4 + i])


if __name__ == '__main__':
    main()
