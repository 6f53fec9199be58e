"""Fine-tune driver for the classification tasks: counterpart of the reference's finetune.py (`get_args_finetune`
:14-72, `FinetuneTrainer` :75-256, `load_data_finetune` :259-330) and main.finetune() (main.py:103-215) on top of the HIP backbone.

Same flags, same batch semantics (decoder input = encoder input for the melody task and for sequence tasks, model.py:203 /
finetune.py:197-198), same loss (masked mean of per-token CE, or mean over the batch for sequence tasks), same accuracy
bookkeeping, checkpoint dict keys and log line. Differences: one process per GPU instead of nn.DataParallel; the backbone is
stepped by the engine's fused HF-AdamW on its flat buffers and the head parameters by the same kernel on a second flat buffer
(the reference builds ONE transformers.AdamW over model.parameters(); it does not clip in fine-tune, finetune.py:227).
The velocity task's decoder label-embedding swap goes through PianoBart.change_decoder_embedding as in the reference."""
import argparse
import os
import shutil

import numpy as np
import torch
from torch.utils.data import Dataset

from . import heads, ops
from ._lib import LIB, PBError
from .model import SequenceClassification, TokenClassification, checkpoint_state_dict


def get_args_finetune(argv=None):
    parser = argparse.ArgumentParser(description='')
    parser.add_argument('--task', choices=['melody', 'velocity', 'composer', 'emotion'], required=True)
    parser.add_argument('--dataset', type=str, choices=('asap', 'Pianist8', 'POP909', 'EMOPIA', 'GiantMIDI1k'), required=True)
    parser.add_argument('--dataroot', type=str, default=None)
    parser.add_argument('--dict_file', type=str, default='./Data/Octuple.pkl')
    parser.add_argument('--name', type=str, default='pianobart')
    parser.add_argument('--ckpt', default='result/pretrain/pianobart/model_best.ckpt')
    parser.add_argument('--num_workers', type=int, default=5)
    parser.add_argument('--class_num', type=int, default=None)
    parser.add_argument('--batch_size', type=int, default=8)
    parser.add_argument('--max_seq_len', type=int, default=1024, help='all sequences are padded to `max_seq_len`')
    parser.add_argument('--hs', type=int, default=1024)
    parser.add_argument('--layers', type=int, default=8)
    parser.add_argument('--ffn_dims', type=int, default=2048)
    parser.add_argument('--heads', type=int, default=8)
    parser.add_argument('--epochs', type=int, default=50, help='number of training epochs')
    parser.add_argument('--lr', type=float, default=2e-5, help='initial learning rate')
    parser.add_argument('--nopretrain', action='store_true')
    parser.add_argument('--cpu', action='store_true')
    parser.add_argument('--cuda_devices', type=int, nargs='+', default=[2, 5, 6], help='CUDA device ids')
    parser.add_argument('--weight', type=float, default=None, help='weight of regularization')
    parser.add_argument('--error_correction', action='store_true')
    parser.add_argument('--precision', choices=['bf16', 'fp32', 'bf16x3'], default='bf16', help='backbone arithmetic (not in the reference)')
    args = parser.parse_args(argv)
    if args.class_num is None:
        args.class_num = {'melody': 4, 'velocity': 7, 'composer': 8, 'emotion': 4}[args.task]
    return args


class FinetuneDataset(Dataset):
    """dataset.py:19-32."""

    def __init__(self, X, y):
        self.data, self.label = X, y

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return torch.tensor(self.data[index]), torch.tensor(self.label[index])


def load_data_finetune(dataset, task, data_root=None):
    """finetune.py:259-330 (the classification branch; `gen` lives in finetune_generation.py)."""
    if data_root is None:
        data_root = 'Data/finetune/others'
    if dataset == 'emotion':
        dataset = 'emopia'
    if dataset not in ['POP909', 'pop909', 'composer', 'EMOPIA', 'asap', 'Pianist8', 'maestro', 'GiantMIDI1k']:
        print(f'Dataset {dataset} not supported')
        exit(1)
    ld = lambda name: np.load(os.path.join(data_root, name), allow_pickle=True)
    X_train, X_val, X_test = ld(f'{dataset}_train.npy'), ld(f'{dataset}_valid.npy'), ld(f'{dataset}_test.npy')
    print('X_train: {}, X_valid: {}, X_test: {}'.format(X_train.shape, X_val.shape, X_test.shape))
    y_train, y_val, y_test = ld(f'{dataset}_train_ans.npy'), ld(f'{dataset}_valid_ans.npy'), ld(f'{dataset}_test_ans.npy')
    print('y_train: {}, y_valid: {}, y_test: {}'.format(y_train.shape, y_val.shape, y_test.shape))
    return X_train, X_val, X_test, y_train, y_val, y_test


class HeadAdamW:
    """HF AdamW (eps added before the bias correction, decay after the update: SURVEY 8a-11) over the head parameters, which are
    re-homed as views of one flat f32 buffer so that a step is one pb_adamw_step launch per run of neighbouring parameters.
    Like transformers.AdamW (`if p.grad is None: continue`) a parameter without a gradient is left alone -- no update, no weight
    decay, its own step count does not advance; parameters in `never` (BART's dead `shared` table) are not re-homed at all."""

    def __init__(self, params, lr, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-6, never=()):
        skip = {id(p) for p in never}
        self.params = [p for p in params if id(p) not in skip]
        dev = self.params[0].device
        n = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        self.P, self.G = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.views, o = [], 0
        for p in self.params:
            v = self.P[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            self.views.append((o, (p.numel() + 3) // 4 * 4))
            o += (p.numel() + 3) // 4 * 4
        self.steps = [0] * len(self.params)                                # per-parameter step counts, as in the HF optimizer state
        self.lr, self.wd, self.betas, self.eps, self.t = lr, weight_decay, betas, eps, 0

    def gather_grads(self):
        """Copy every parameter's .grad into the flat gradient buffer (what a data-parallel caller all-reduces before step)."""
        for p, (o, n) in zip(self.params, self.views):
            if p.grad is not None:
                self.G[o:o + p.numel()].copy_(p.grad.reshape(-1))
            else:
                self.G[o:o + n].zero_()

    def step(self, gathered=False):
        self.t += 1
        runs = []                                                          # (offset, length, step): neighbours with a gradient and equal step count
        for i, (p, (o, n)) in enumerate(zip(self.params, self.views)):
            if p.grad is None:
                continue
            if not gathered:
                self.G[o:o + p.numel()].copy_(p.grad.reshape(-1))
            self.steps[i] += 1
            if runs and runs[-1][0] + runs[-1][1] == o and runs[-1][2] == self.steps[i]:
                runs[-1] = (runs[-1][0], runs[-1][1] + n, self.steps[i])
            else:
                runs.append((o, n, self.steps[i]))
        for o, n, t in runs:
            ops.adamw_step(self.P[o:o + n], self.G[o:o + n], self.m[o:o + n], self.v[o:o + n], None, None, self.lr, self.betas[0],
                           self.betas[1], self.eps, self.wd, t)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class FinetuneTrainer:
    def __init__(self, pianobart, train_dataloader, valid_dataloader, test_dataloader, lr, class_num, hs, testset_shape, cpu,
                 cuda_devices=None, model=None, SeqClass=False, error=False, weight=None, data_parallel=None):
        if cpu or not torch.cuda.is_available():
            raise PBError('pianobart_amd has no CPU execution path')
        if cuda_devices is not None and len(cuda_devices) > 1:
            raise PBError('nn.DataParallel is replaced by one process per GPU (torch.distributed.run)')
        dev_id = int(os.environ['LOCAL_RANK']) if 'LOCAL_RANK' in os.environ else (cuda_devices[0] if cuda_devices else 0)
        self.device = torch.device('cuda', dev_id)
        torch.cuda.set_device(self.device)
        print('   device:', self.device)
        self.pianobart, self.SeqClass, self.class_num = pianobart, SeqClass, class_num
        if model is not None:
            print('load a fine-tuned model')
            self.model = model.to(self.device)
        else:
            print('init a fine-tune model, sequence-level task?', SeqClass)
            self.model = (SequenceClassification(pianobart, class_num, hs) if SeqClass else TokenClassification(pianobart, class_num + 1, hs)).to(self.device)
        print('Use GPU', self.device)
        self.engine = pianobart._get_engine()
        self.engine.bind(self.device)
        self.train_data, self.valid_data, self.test_data = train_dataloader, valid_dataloader, test_dataloader
        in_engine = {id(p) for p in self.engine.params}              # everything else (heads, swapped decoder label embedding) is stepped here
        self.head_optim = HeadAdamW([p for p in self.model.parameters() if id(p) not in in_engine], lr=lr, weight_decay=0.01,
                                    never=[pianobart.bart.shared.weight])      # never read, never a gradient (SURVEY a-3)
        self.lr = lr
        # one process per GPU (torch.distributed.run): every rank steps its own shard of the global mini-batch (make_finetune_loaders),
        # gradients are AVERAGED over the ranks -- the
        # backbone's through the engine's bucket exchange (summed, then scaled by 1 / world in the optimizer step), the head parameters'
        # with one all-reduce of the head optimizer's flat gradient buffer. data_parallel=True installs this at world size 1 (tests).
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.reducer = None
        if self.world > 1 or data_parallel:
            import torch.distributed as dist
            from .parallel import GradReducer
            if not dist.is_initialized():
                dist.init_process_group('nccl', device_id=self.device)
            self.reducer = GradReducer(self.engine, self.world)
        self.testset_shape = testset_shape if not error else testset_shape[:-1]
        self.error = error
        self.weight = weight
        if weight is not None:
            self._l2_scratch = torch.empty(int(LIB.query('pb_l2_penalty_scratch_floats')), dtype=torch.float32, device=self.device)
            self._l2_acc = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._engine_slot = {id(p): i for i, p in enumerate(self.engine.params)}

    def l2_penalty(self, with_grad):
        """finetune.py:241-243: `loss += weight * torch.norm(param, p=2)` over model.parameters(). Returns the penalty (device scalar);
        with_grad adds weight * p / ||p|| to every gradient (for the backbone: in the engine's flat gradient buffer the optimizer reads).
        BART's never-read `shared` table has no gradient here (SURVEY a-3): it only contributes its norm to the reported loss."""
        ops.fill_f32(self._l2_acc, 0.0)
        for p in self.model.parameters():
            if not p.is_cuda or p.dtype != torch.float32 or not p.data.is_contiguous():
                raise PBError('l2_penalty: parameters must be contiguous f32 HIP tensors')
            g = None
            i = self._engine_slot.get(id(p))
            if with_grad:
                g = self.engine.grad_views[i] if i is not None else p.grad
            # data parallel: the backbone's flat gradient holds the SUM over ranks and is scaled by 1 / world in the optimizer step,
            # so the (rank-independent) penalty gradient enters it world times; the reported value is not affected
            if with_grad and i is not None and self.world > 1:
                ops.l2_penalty(p.data, None, self.weight, self._l2_scratch, self._l2_acc)
                ops.l2_penalty(p.data, g, self.weight * self.world, self._l2_scratch, None)
            else:
                ops.l2_penalty(p.data, g, self.weight, self._l2_scratch, self._l2_acc)
        return self._l2_acc

    def compute_loss(self, predict, target, loss_mask, seq):
        """finetune.py:121-129; predict (..., C) (the reference permutes to (B, C, S) for nn.CrossEntropyLoss)."""
        loss = heads.cross_entropy_rows(predict, target)
        if not seq:
            return torch.sum(loss * loss_mask) / torch.sum(loss_mask)
        return torch.sum(loss) / loss.shape[0]

    def train(self):
        self.model.train()
        return self.iteration(self.train_data, 0, self.SeqClass)

    def valid(self):
        self.model.eval()
        return self.iteration(self.valid_data, 1, self.SeqClass)

    def test(self):
        self.model.eval()
        return self.iteration(self.test_data, 2, self.SeqClass)

    def iteration(self, training_data, mode, seq):
        total_acc, total_cnt, total_loss = 0.0, 0, 0.0
        self.model.train(mode == 0)
        sampler = getattr(training_data, 'sampler', None)
        if mode == 0 and hasattr(sampler, 'set_epoch'):                      # rank-sharded training set: a new permutation per epoch
            self._epoch = getattr(self, '_epoch', -1) + 1
            sampler.set_epoch(self._epoch)
        all_output, cnt = (torch.empty(self.testset_shape) if mode == 2 else None), 0
        with torch.set_grad_enabled(mode == 0):
            for x, y in training_data:
                batch = x.shape[0]
                x, y = x.to(self.device).long(), y.to(self.device).long()
                if self.error:
                    y = torch.squeeze(y, dim=-1)
                attn = (x[:, :, 0] != self.pianobart.bar_pad_word).float()
                if seq:
                    y_hat = self.model(input_ids_encoder=x, encoder_attention_mask=attn)
                else:
                    if self.class_num >= 5:                              # velocity: labels shifted right, class_num = the pad label
                        y_shift = torch.zeros_like(y) + self.class_num
                        y_shift[:, 1:] = y[:, :-1]
                        attn_shift = torch.zeros_like(attn)
                        attn_shift[:, 1:] = attn[:, :-1]
                        attn_shift[:, 0] = attn[:, 0]
                    else:
                        y_shift, attn_shift = x, attn
                    y_hat = self.model(input_ids_encoder=x, input_ids_decoder=y_shift, encoder_attention_mask=attn, decoder_attention_mask=attn_shift)
                output = torch.from_numpy(np.argmax(y_hat.detach().cpu().numpy(), axis=-1)).to(self.device)
                if mode == 2:
                    all_output[cnt:cnt + batch] = output.cpu()
                    cnt += batch
                if not seq:
                    total_acc += float(torch.sum((y == output).float() * attn))
                    total_cnt += float(torch.sum(attn))
                else:
                    total_acc += float(torch.sum((y == output).float()))
                    total_cnt += y.shape[0]
                loss = self.compute_loss(y_hat, y, attn, seq)
                total_loss += float(loss.detach())
                if mode == 0:
                    self.model.zero_grad()
                    self.head_optim.zero_grad()
                    loss.backward()
                    if self.reducer:
                        self.reducer.all_reduce_grads()                      # the backbone buckets have been exchanged (summed) before anything else touches them
                if self.weight is not None:                                  # the regulariser is part of the reported loss in every mode
                    total_loss += float(self.l2_penalty(mode == 0))
                if mode == 0:
                    if self.reducer:
                        import torch.distributed as dist
                        self.head_optim.gather_grads()
                        dist.all_reduce(self.head_optim.G)
                        self.head_optim.G.mul_(1.0 / self.world)
                    self.engine.optimizer_step(lr=self.lr, max_norm=float('inf'), gscale=1.0 / self.world)   # no clipping in fine-tune (finetune.py:227)
                    self.head_optim.step(gathered=self.reducer is not None)
        nb = len(training_data)
        if mode == 0 and self.world > 1:                                     # every rank saw its own shard: report the job's numbers
            total_loss, total_acc, total_cnt, nb = reduce_epoch_sums([total_loss, total_acc, total_cnt, nb], self.device)
        res = (round(total_loss / nb, 4), round(total_acc / total_cnt, 4))
        return res + (all_output,) if mode == 2 else res

    def save_checkpoint(self, epoch, train_acc, valid_acc, valid_loss, train_loss, is_best, filename):
        state = {'epoch': epoch + 1, 'state_dict': self.model.state_dict(), 'valid_acc': valid_acc, 'valid_loss': valid_loss,
                 'train_loss': train_loss, 'train_acc': train_acc,
                 'optimizer': {'backbone_step': self.engine.step_count, 'head_step': self.head_optim.t}}
        torch.save(state, filename)
        if is_best:
            shutil.copyfile(filename, filename.split('.')[0] + '_best.ckpt')


def make_finetune_loaders(arrays, batch_size, num_workers, seed=2023):
    """The three DataLoaders of main.py:126-141 / 240-255 (train shuffled) from (X_train, X_val, X_test, y_train, y_val, y_test).
    Under torchrun `batch_size` stays the GLOBAL batch (what nn.DataParallel scatters): the TRAIN loader hands each rank
    batch_size / world samples per step through a DistributedSampler (same permutation seed on every rank; the trainer calls
    set_epoch); the validation and test loaders stay whole on every rank, so every rank reports the same numbers and the
    (N, S) test output needs no gather."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from .pretrain import _dist_env, _loader_kw                               # persistent workers (see there)
    X_train, X_val, X_test, y_train, y_val, y_test = arrays
    rank, world = _dist_env()
    kw = _loader_kw(num_workers)
    train_ds = FinetuneDataset(X=X_train, y=y_train)
    if world > 1:
        if batch_size % world:
            raise PBError('--batch_size %d is the global batch and must be a multiple of the %d ranks' % (batch_size, world))
        sampler = DistributedSampler(train_ds, num_replicas=world, rank=rank, shuffle=True, seed=seed)
        train = DataLoader(train_ds, batch_size=batch_size // world, sampler=sampler, **kw)
    else:
        train = DataLoader(train_ds, batch_size=batch_size, shuffle=True, **kw)
    return [train] + [DataLoader(FinetuneDataset(X=X, y=y), batch_size=batch_size, **kw) for X, y in ((X_val, y_val), (X_test, y_test))]


def reduce_epoch_sums(values, device):
    """Sum a list of per-rank epoch counters over the ranks (training metrics of a rank-sharded epoch); identity at world size 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t)
    return t.cpu().tolist()


def _finetune_loaders(args):
    """The three DataLoaders of main.py:126-141 (train shuffled; rank-sharded under torchrun) + the shape of the test labels."""
    X_train, X_val, X_test, y_train, y_val, y_test = load_data_finetune(args.dataset, args.task, args.dataroot)
    loaders = make_finetune_loaders((X_train, X_val, X_test, y_train, y_val, y_test), args.batch_size, args.num_workers)
    for tag, ld in zip(('train', 'valid', 'valid'), loaders):                  # the reference prints "valid_loader" twice
        print('   len of %s_loader' % tag, len(ld))
    return loaders, y_test.shape


def finetune(argv=None):
    """The classification fine-tune driver (reference: main.py:103-215): seeds 2023, the reference's flags, prints, checkpoint keys
    and log lines; best model = highest validation accuracy (ties count), stop after 4 epochs without one."""
    import random
    from .model import BartConfig, PianoBart
    from .pretrain import _load_vocab
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(2023)
    args = get_args_finetune(argv)
    print('Loading Dictionary')
    e2w, w2e = _load_vocab(args.dict_file)
    print('\nLoading Dataset')
    (train_loader, valid_loader, test_loader), test_shape = _finetune_loaders(args)
    print('\nBuilding BART model')
    shape = dict(max_position_embeddings=args.max_seq_len, d_model=args.hs)
    for side in ('encoder', 'decoder'):
        shape.update({side + '_layers': args.layers, side + '_ffn_dim': args.ffn_dims, side + '_attention_heads': args.heads})
    pianobart = PianoBart(bartConfig=BartConfig(**shape), e2w=e2w, w2e=w2e, precision=args.precision)
    best_mdl = '' if args.nopretrain else args.ckpt
    if best_mdl:
        print('   Loading pre-trained model from', best_mdl.split('/')[-1])
        pianobart.load_state_dict(checkpoint_state_dict(torch.load(best_mdl, map_location='cpu', weights_only=False)['state_dict']))
    print('\nCreating Finetune Trainer')
    trainer = FinetuneTrainer(pianobart, train_loader, valid_loader, test_loader, args.lr, args.class_num, args.hs, test_shape, args.cpu,
                              args.cuda_devices[:1], None, args.task in ('composer', 'emotion'), args.error_correction, args.weight)
    print('\nTraining Start')
    save_dir = os.path.join('result/finetune/', args.task + '_' + args.name)
    os.makedirs(save_dir, exist_ok=True)
    filename = os.path.join(save_dir, 'model.ckpt')
    print('   save model at {}'.format(filename))
    best_acc, stale = 0, 0
    rank0 = int(os.environ.get('RANK', 0)) == 0                                # one writer: every rank holds the same model and numbers
    with open(os.path.join(save_dir, 'log') if rank0 else os.devnull, 'a') as log:
        log.write('Loading pre-trained model from ' + best_mdl.split('/')[-1] + '\n')
        for epoch in range(args.epochs):
            (train_loss, train_acc), (valid_loss, valid_acc) = trainer.train(), trainer.valid()
            test_loss, test_acc, _ = trainer.test()
            is_best = valid_acc >= best_acc
            best_acc = max(valid_acc, best_acc)
            stale = 0 if is_best else stale + 1
            print('epoch: {}/{} | Train Loss: {} | Train acc: {} | Valid Loss: {} | Valid acc: {} | Test loss: {} | Test acc: {}'.format(
                epoch + 1, args.epochs, train_loss, train_acc, valid_loss, valid_acc, test_loss, test_acc))
            if rank0:
                trainer.save_checkpoint(epoch, train_acc, valid_acc, valid_loss, train_loss, is_best, filename)
            log.write('Epoch {}: train_loss={}, valid_loss={}, test_loss={}, train_acc={}, valid_acc={}, test_acc={}\n'.format(
                epoch + 1, train_loss, valid_loss, test_loss, train_acc, valid_acc, test_acc))
            log.flush()
            if stale > 3:
                print('valid acc not improving for 3 epochs')
                break
    return trainer


if __name__ == '__main__':
    finetune()
