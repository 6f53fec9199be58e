"""Teacher-forced generation fine-tune on MI355X: counterpart of the reference's `GenerationTrainer`
(finetune_generation.py:57-290) -- SURVEY 8(f-2), the first "next" row after the pre-train path.

Same forward/backward as the pre-train step with a different mask and weights:
  * decoder input = the encoder input itself (`y_shift = x`, finetune_generation.py:155), masks = bar column != PAD;
  * per-head CE averaged over the decoder attention mask, head weights 0.3 (Instrument, TimeSig, Tempo) / 1.5 (Pitch) / 1,
    times len(e2w[etype]) in dict order, divided by sum(n_tok) (finetune_generation.py:236-250);
  * accuracy over the same mask (finetune_generation.py:188-193); argmax ids collected in test mode.
The shape-similarity "FAD" metrics (finetune_generation.py:185-223) need the third-party `shapesimilarity` package, which is
not installed in this image (a host-side metric, outside the hot path): they are reported as `n/a` / None, never as a number.
`get_args_generation` and `finetune_generation()` are the counterparts of finetune_generation.py:15-55 and main.py:214-321.
"""
import argparse
import os
import shutil
import sys

import numpy as np
import torch

from . import ops
from ._lib import PBError
from .model import PianoBartLM, checkpoint_state_dict

HEAD_WEIGHT = [1.0, 1.0, 0.3, 1.5, 1.0, 1.0, 0.3, 0.3]          # finetune_generation.py:241-248 (index = head i)


def get_args_generation(argv=None):
    """finetune_generation.py:15-55, flag for flag (+ --precision, absent from the reference)."""
    parser = argparse.ArgumentParser(description='')
    parser.add_argument("--datasets", type=str, default='maestro')
    parser.add_argument('--dict_file', type=str, default='./Data/Octuple.pkl')
    parser.add_argument('--name', type=str, default='pianobart')
    parser.add_argument('--ckpt', default='result/pretrain/pianobart/model_best.ckpt')
    parser.add_argument('--num_workers', type=int, default=5)
    parser.add_argument('--batch_size', type=int, default=8)
    parser.add_argument('--max_seq_len', type=int, default=1024, help='all sequences are padded to `max_seq_len`')
    parser.add_argument('--hs', type=int, default=1024)
    parser.add_argument('--layers', type=int, default=8)
    parser.add_argument('--ffn_dims', type=int, default=2048)
    parser.add_argument('--heads', type=int, default=8)
    parser.add_argument('--epochs', type=int, default=500, help='number of training epochs')
    parser.add_argument('--lr', type=float, default=2e-6, help='initial learning rate')
    parser.add_argument('--nopretrain', action="store_true")
    parser.add_argument('--dataroot', type=str, default=None, help='path to dataset')
    parser.add_argument("--cpu", action="store_true")
    parser.add_argument("--cuda_devices", type=int, nargs='+', default=[0], help="HIP device ids (one per process)")
    parser.add_argument("--eval", action="store_true")
    parser.add_argument('--precision', choices=['bf16', 'fp32', 'bf16x3'], default='bf16', help='backbone arithmetic (not in the reference)')
    return parser.parse_args(argv)


def load_data_generation(dataset, data_root=None):
    """finetune.py:259-330, task "gen": <root>/<dataset>_{train,valid,test}.npy and ..._ans.npy (prompt / continuation pairs)."""
    if data_root is None:
        data_root = 'Data/finetune/gen'
    ld = lambda name: np.load(os.path.join(data_root, name), allow_pickle=True)
    X = [ld(f'{dataset}_{part}.npy') for part in ('train', 'valid', 'test')]
    y = [ld(f'{dataset}_{part}_ans.npy') for part in ('train', 'valid', 'test')]
    print('X_train: {}, X_valid: {}, X_test: {}'.format(*[a.shape for a in X]))
    print('y_train: {}, y_valid: {}, y_test: {}'.format(*[a.shape for a in y]))
    return X[0], X[1], X[2], y[0], y[1], y[2]


class GenerationTrainer:
    def __init__(self, pianobart, train_dataloader, valid_dataloader, test_dataloader, lr, testset_shape, cpu, cuda_devices=None, model=None):
        if cpu or not torch.cuda.is_available():
            raise PBError('pianobart_amd has no CPU execution path')
        if cuda_devices is not None and len(cuda_devices) > 1:
            raise PBError('nn.DataParallel is replaced by one process per GPU (torch.distributed.run)')
        dev_id = int(os.environ['LOCAL_RANK']) if 'LOCAL_RANK' in os.environ else (cuda_devices[0] if cuda_devices else 0)
        self.device = torch.device('cuda', dev_id)                              # torchrun: one GPU per rank
        torch.cuda.set_device(self.device)
        print('   device:', self.device)
        self.pianobart = pianobart
        self.model = (model if model is not None else PianoBartLM(pianobart)).to(self.device)
        self.engine = self.model._get_engine()
        self.engine.bind(self.device)
        self.train_data, self.valid_data, self.test_data = train_dataloader, valid_dataloader, test_dataloader
        self.testset_shape = testset_shape
        self.lr = lr
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.reducer = None
        if self.world > 1:                                                   # one process per GPU, same exchange as the pre-train step
            import torch.distributed as dist
            from .parallel import GradReducer
            if not dist.is_initialized():
                dist.init_process_group('nccl', device_id=self.device)
            self.reducer = GradReducer(self.engine, self.world)
        n_tok = np.array([len(pianobart.e2w[k]) for k in pianobart.e2w], dtype=np.float64)       # dict order, as the reference
        w = np.array(HEAD_WEIGHT) * n_tok
        self._hw = torch.tensor(w, dtype=torch.float32, device=self.device)
        self._scale = float(w.sum() / n_tok.sum())
        self._wnp, self._ntok = np.array(HEAD_WEIGHT), n_tok

    def train(self):
        self.model.train()
        return self.iteration(self.train_data, 0)

    def valid(self):
        self.model.eval()
        return self.iteration(self.valid_data, 1)

    def test(self):
        self.model.eval()
        return self.iteration(self.test_data, 2)

    def iteration(self, training_data, mode):
        eng, pad = self.engine, int(self.pianobart.bar_pad_word)
        total_acc, total_loss = np.zeros(8), 0.0
        all_output, cnt = (torch.empty(self.testset_shape) if mode == 2 else None), 0
        sampler = getattr(training_data, 'sampler', None)
        if mode == 0 and hasattr(sampler, 'set_epoch'):                      # rank-sharded training set (make_finetune_loaders)
            self._epoch = getattr(self, '_epoch', -1) + 1
            sampler.set_epoch(self._epoch)
        for x, y in training_data:
            x, y = x.to(self.device).long(), y.to(self.device).long()
            B, S = x.shape[:2]
            x16, y16 = ops.ids_to_i16(x), ops.ids_to_i16(y)
            attn_enc = (x16[:, :, 0] != pad).float()
            attn_dec = attn_enc                                        # y_shift = x
            loss_mask = attn_dec[:, :, None].expand(B, S, 8).contiguous()
            am = torch.empty(B * S, 8, dtype=torch.int16, device=self.device) if mode == 2 else None
            sums = eng.loss_and_grads(x16, x16, y16, loss_mask, attn_enc, attn_dec, train=(mode == 0), head_w=self._hw, w_scale=self._scale,
                                      argmax_out=am, count_hook=self.reducer.reduce_counts if self.reducer else None)
            if mode == 0:
                if self.reducer:
                    self.reducer.all_reduce_grads()
                eng.optimizer_step(lr=self.lr)
            if self.reducer:
                self.reducer.reduce_sums(sums)
            s = sums.double().cpu().numpy()
            losses = s[0:8] / s[8:16] * self._wnp                      # the reference logs the weighted per-head losses
            accs = s[16:24] / s[8:16]
            loss = float((losses * self._ntok).sum() / self._ntok.sum())
            if mode == 2:
                all_output[cnt:cnt + B] = am.view(B, S, 8).float().cpu()
                cnt += B
            sys.stdout.write('Loss: {:06f} | loss: {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}\n'.format(loss, *losses))
            sys.stdout.write('Acc: {:06f} | acc: {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}\n'.format(np.average(accs), *accs))
            sys.stdout.write('FAD(BAR) Similarity: n/a , FAD Similarity n/a \n')            # `shapesimilarity` is not installed: not measured
            total_acc += accs
            total_loss += loss
        n = max(1, len(training_data))
        out = (round(total_loss / n, 4), [round(float(a) / n, 4) for a in total_acc], None, None)           # FAD(BAR), FAD: not measured
        return out + (all_output,) if mode == 2 else out

    def save_checkpoint(self, epoch, train_acc, valid_acc, valid_loss, train_loss, is_best, filename):
        """finetune_generation.py:276-290: whole-model state_dict (pianobart.* + mask_lm.*)."""
        eng = self.engine
        state = {'epoch': epoch + 1, 'state_dict': {k: v.detach().cpu() for k, v in self.model.state_dict().items()}, 'valid_acc': valid_acc,
                 'valid_loss': valid_loss, 'train_loss': train_loss, 'train_acc': train_acc,
                 'optimizer': dict(eng.optimizer_state(self.model), lr=self.lr)}            # moments keyed by parameter name
        torch.save(state, filename)
        if is_best:
            shutil.copyfile(filename, filename.split('.')[0] + '_best.ckpt')


def finetune_generation(argv=None):
    """The generation fine-tune driver (reference: main.py:214-321): seeds 2023, maestro-style (prompt, continuation) arrays, a
    pre-trained PianoBart checkpoint (or a whole fine-tuned PianoBartLM with --eval), per epoch train -> valid -> test, best model by
    the n_tokens-weighted validation accuracy, the reference's checkpoint keys and log / stdout line formats."""
    import pickle
    import random
    from .finetune import make_finetune_loaders
    from .model import BartConfig, PianoBart
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(2023)
    args = get_args_generation(argv)
    print("Loading Dictionary")
    if args.dict_file.endswith('.json'):
        import json
        e2w = json.load(open(args.dict_file))['e2w']
        w2e = {k: {v: w for w, v in d.items()} for k, d in e2w.items()}
    else:
        with open(args.dict_file, 'rb') as f:
            e2w, w2e = pickle.load(f)
    print("\nLoading Dataset")
    X_train, X_val, X_test, y_train, y_val, y_test = load_data_generation(args.datasets, args.dataroot)
    # train loader rank-sharded under torchrun (global --batch_size), validation / test whole on every rank
    loaders = make_finetune_loaders((X_train, X_val, X_test, y_train, y_val, y_test), args.batch_size, args.num_workers)
    for tag, ld in zip(('train', 'valid', 'valid'), loaders):
        print("   len of %s_loader" % tag, len(ld))
    print("\nBuilding BART model")
    pianobart = PianoBart(bartConfig=BartConfig(max_position_embeddings=args.max_seq_len, d_model=args.hs, encoder_layers=args.layers,
                                                encoder_ffn_dim=args.ffn_dims, encoder_attention_heads=args.heads, decoder_layers=args.layers,
                                                decoder_ffn_dim=args.ffn_dims, decoder_attention_heads=args.heads),
                          e2w=e2w, w2e=w2e, precision=args.precision)
    best_mdl, model = '', None
    if args.eval or not args.nopretrain:
        best_mdl = args.ckpt
        print("   Loading pre-trained model from", best_mdl.split('/')[-1])
        sd = checkpoint_state_dict(torch.load(best_mdl, map_location='cpu', weights_only=False)['state_dict'])
        if args.eval:                                                          # a whole fine-tuned PianoBartLM
            model = PianoBartLM(pianobart)
            model.load_state_dict(sd)
        else:
            pianobart.load_state_dict(sd)
    print("\nCreating Finetune Trainer")
    trainer = GenerationTrainer(pianobart, loaders[0], loaders[1], loaders[2], args.lr, y_test.shape, args.cpu, args.cuda_devices, model)
    print("\nTraining Start")
    save_dir = os.path.join('result/finetune/generation_' + args.name)
    os.makedirs(save_dir, exist_ok=True)
    filename = os.path.join(save_dir, 'model.ckpt')
    print("   save model at {}".format(filename))
    rank0 = int(os.environ.get('RANK', 0)) == 0
    best_acc, bad_cnt = 0, 0
    log = open(os.path.join(save_dir, 'log'), 'a') if rank0 else None
    if log:
        log.write("Loading pre-trained model from " + best_mdl.split('/')[-1] + '\n')
    for epoch in range(args.epochs):
        res = {part: getattr(trainer, part)() for part in ('train', 'valid', 'test')}
        (train_loss, train_acc, train_fb, train_f), (valid_loss, valid_acc, valid_fb, valid_f) = res['train'], res['valid']
        test_loss, test_acc, test_fb, test_f, _ = res['test']
        avg_acc = sum(a * n for a, n in zip(valid_acc, pianobart.n_tokens)) / sum(pianobart.n_tokens)
        is_best = avg_acc > best_acc
        best_acc = max(avg_acc, best_acc)
        bad_cnt = 0 if is_best else bad_cnt + 1
        print('epoch: {}/{} | Train Loss: {} | Train acc: {} | Train FAD: {} | Train FAD (BAR): {} | Valid Loss: {} | Valid acc: {} | Valid FAD: {} | '
              'Valid FAD(BAR): {} | Test loss: {} | Test acc: {} | Test FAD: {} | Test FAD(BAR): {}'.format(
                  epoch + 1, args.epochs, train_loss, train_acc, train_f, train_fb, valid_loss, valid_acc, valid_f, valid_fb, test_loss, test_acc,
                  test_f, test_fb))
        if log:
            trainer.save_checkpoint(epoch, train_acc, valid_acc, valid_loss, train_loss, is_best, filename)
            log.write('Epoch {}: train_loss={}, valid_loss={}, test_loss={}, train_acc={}, valid_acc={}, test_acc={}, train_fad={}, valid_fad={}, '
                      'test_fad={}, train_fad(bar)={}, valid_fad(bar)={}, test_fad(bar)={}\n'.format(
                          epoch + 1, train_loss, valid_loss, test_loss, train_acc, valid_acc, test_acc, train_f, valid_f, test_f, train_fb, valid_fb, test_fb))
            log.flush()
        if bad_cnt > 30:
            print('valid acc not improving for 3 epochs')
            break
    if log:
        log.close()
    return trainer


if __name__ == '__main__':
    finetune_generation()
