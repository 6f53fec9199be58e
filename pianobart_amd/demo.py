"""MIDI in -> PianoBART continuation -> MIDI out: the reference's demo.py:105-170 on the HIP path.

Same `Args` fields and command-line flags as the reference (demo.py:12-29, 33-58), same order of work: vocabulary, model from the
flags, checkpoint with `load_state_dict(strict=False)` (demo.py:128-129), `Midi2Octuple` (demo.py:61-68), encoder mask = bar
column != PAD, `model(generate=True)` (demo.py:157; here one decoder token per step against the K/V caches instead of a full
encoder + decoder pass per position: same tokens), `Octuple2Midi` (demo.py:72-102). No CPU path (`--cpu` raises).
"""
import argparse
import os

import torch

from ._lib import PBError
from .model import BartConfig, PianoBart, PianoBartLM, checkpoint_state_dict
from .octuple_midi import Midi2Octuple, Octuple2Midi

_HERE = os.path.dirname(os.path.abspath(__file__))
_VOCAB = os.path.join(_HERE, 'data', 'octuple_vocab.json')


class Args:
    """demo.py:12-29: what gui/backend/app.py builds instead of a command line."""

    def __init__(self, dict_file=_VOCAB, ckpt='./PianoBART_Giant.ckpt', input='./Data/POP909/POP909/001/001.mid', output='./output.mid',
                 num_workers=5, max_seq_len=1024, hs=1024, layers=8, ffn_dims=2048, heads=8, nopretrain=False, cpu=False, cuda_devices=[0],
                 precision='bf16'):
        self.dict_file, self.ckpt, self.input, self.output, self.num_workers = dict_file, ckpt, input, output, num_workers
        self.max_seq_len, self.hs, self.layers, self.ffn_dims, self.heads = max_seq_len, hs, layers, ffn_dims, heads
        self.nopretrain, self.cpu, self.cuda_devices, self.precision = nopretrain, cpu, cuda_devices, precision


def get_args(argv=None):
    ap = argparse.ArgumentParser(description='')
    ap.add_argument('--dict_file', type=str, default=_VOCAB)
    ap.add_argument('--ckpt', default='result/pretrain/pianobart/model_best.ckpt')
    ap.add_argument('--input', default='./Data/POP909/POP909/001/001.mid')
    ap.add_argument('--output', default='./output.mid')
    ap.add_argument('--num_workers', type=int, default=5)
    ap.add_argument('--max_seq_len', type=int, default=1024, help='all sequences are padded to `max_seq_len`')
    ap.add_argument('--hs', type=int, default=1024)
    ap.add_argument('--layers', type=int, default=8)
    ap.add_argument('--ffn_dims', type=int, default=2048)
    ap.add_argument('--heads', type=int, default=8)
    ap.add_argument('--nopretrain', action='store_true', default=False)
    ap.add_argument('--cpu', action='store_true')
    ap.add_argument('--cuda_devices', type=int, nargs='+', default=[0], help='HIP device ids (one: generate is batch-1 sequential)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'bf16x3'])
    return ap.parse_args(argv)


def demo(args=None):
    if not args:
        args = get_args()
    if args.cpu or not torch.cuda.is_available():
        raise PBError('pianobart_amd has no CPU execution path: demo() needs an MI355X')
    if args.cuda_devices is not None and len(args.cuda_devices) > 1:
        raise PBError('generate is batch-1 sequential: give ONE device (the reference itself is single-device here, README.md:154)')
    from .pretrain import _load_vocab
    print("Loading Dictionary")
    e2w, w2e = _load_vocab(args.dict_file)
    print("\nBuilding BART model")
    shape = dict(max_position_embeddings=args.max_seq_len, d_model=args.hs)
    for side in ('encoder', 'decoder'):
        shape.update({side + '_layers': args.layers, side + '_ffn_dim': args.ffn_dims, side + '_attention_heads': args.heads})
    pianobart = PianoBart(bartConfig=BartConfig(**shape), e2w=e2w, w2e=w2e, precision=getattr(args, 'precision', 'bf16'))
    model = PianoBartLM(pianobart)
    if not args.nopretrain:
        print("   Loading pre-trained model from", args.ckpt.split('/')[-1])
        sd = torch.load(args.ckpt, map_location='cpu', weights_only=False)['state_dict']
        model.load_state_dict(checkpoint_state_dict(sd, model), strict=False)       # 'module.'-prefixed multi-GPU files load too
    octuple = Midi2Octuple(args.input, window=args.max_seq_len)
    device_num = args.cuda_devices[0] if args.cuda_devices else 0
    device = torch.device('cuda', device_num)
    print("Use GPU", device)
    model = model.to(device).eval()
    octuple = octuple.long().to(device)
    attn_encoder = (octuple[:, :, 0] != pianobart.bar_pad_word).float()
    with torch.no_grad():
        y = model(input_ids_encoder=octuple, encoder_attention_mask=attn_encoder, generate=True, device_num=device_num)
    if Octuple2Midi(y, args.output):
        print(f"Saved to {args.output}")
    print(octuple.shape, y.shape)
    return octuple, y


if __name__ == '__main__':
    demo()
