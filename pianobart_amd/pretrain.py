"""Pre-training harness on MI355X: the counterpart of the reference's `pretrain.py` (Pretrainer,
get_args_pretrain, load_data_pretrain) and `main.pretrain()` (main.py:17-100).

Same CLI flags and defaults (pretrain.py:18-48), same BartConfig field mapping (main.py:39-47), same
per-epoch train -> valid flow, weighted avg-acc best tracking (main.py:72-82), checkpoint dict keys and
file names (pretrain.py:96-110), log line formats (pretrain.py:199-204, main.py:84-92). What differs is
where the work runs: the reference's per-sample Python `gen_mask` + 8 logits D2H copies + >= 17 host syncs
per step become: one corruption kernel per batch, the fused HIP step (Engine.loss_and_grads +
optimizer_step), and ONE device->host copy of 24 floats per step for the log lines.

Multi-GPU: launch one process per GPU (torch.distributed.run); `--cuda_devices` keeps its meaning of
"which device(s)" for a single process, but more than one id is rejected with a pointer to torchrun,
because nn.DataParallel (pretrain.py:63-65) is exactly what this engine replaces. nn.DataParallel scatters
ONE batch of `--batch_size` samples over the GPUs; here every rank draws its `batch_size / world` share of
the same global batch: the shuffle + 85/15 split is drawn on rank 0 and broadcast, each epoch's order is a
permutation seeded identically on all ranks, rank r takes every world-th sample of it (DistributedSampler),
and the corruption / dropout streams are keyed by the rank so that no two ranks draw the same decisions.

Input: the `.npy` files stay memory-mapped (pianobart_amd/data.py); concatenate, shuffle and split act on an
index, and a batch reaches the device as int16 rows (16 bytes per token) -- what the kernels read.
"""
import argparse
import os
import random
import shutil
import sys
import time

import numpy as np
import torch

from . import ops
from ._lib import PBError
from .data import BalancedDistributedSampler, MidiDataset, OctupleShards, sequence_lengths
from .model import BartConfig, PianoBart, PianoBartLM


def get_args_pretrain(argv=None):
    """pretrain.py:18-48, flag for flag."""
    parser = argparse.ArgumentParser(description='')
    parser.add_argument('--dict_file', type=str, default='./Data/Octuple.pkl')
    parser.add_argument('--name', type=str, default='pianobart')
    parser.add_argument("--datasets", type=str, nargs='+', default=['asap', 'EMOPIA', 'Pianist8', 'POP1K7', 'POP909'])
    parser.add_argument('--num_workers', type=int, default=5)
    parser.add_argument('--batch_size', type=int, default=16)
    parser.add_argument('--mask_percent', type=float, default=0.15,
                        help="Up to `valid_seq_len * target_max_percent` tokens will be masked out for prediction")
    parser.add_argument('--max_seq_len', type=int, default=1024, help='all sequences are padded to `max_seq_len`')
    parser.add_argument('--hs', type=int, default=1024)
    parser.add_argument('--layers', type=int, default=8)
    parser.add_argument('--ffn_dims', type=int, default=2048)
    parser.add_argument('--heads', type=int, default=8)
    parser.add_argument('--epochs', type=int, default=500, help='number of training epochs')
    parser.add_argument('--lr', type=float, default=2e-5, help='initial learning rate')
    parser.add_argument("--cpu", action="store_true")
    parser.add_argument("--cuda_devices", type=int, nargs='+', default=[0], help="HIP device ids (one per process)")
    # build-only additions (absent from the reference)
    parser.add_argument('--precision', type=str, default='bf16', choices=['bf16', 'fp32', 'bf16x3'])
    parser.add_argument('--data_root', type=str, default='Data/output_pretrain')
    parser.add_argument('--quiet', action='store_true', help='do not print the two per-step Loss/Acc lines')
    parser.add_argument('--resume', type=str, default='', help='continue from a checkpoint THIS driver wrote: weights, LM heads, AdamW moments and step, epoch, best_acc')
    return parser.parse_args(argv)


def _dist_env():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def _dist_init(device=None):
    """One process per GPU: join the job's process group (RCCL on the HIP device; gloo when there is none, i.e. the CPU tests of
    the input pipeline)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        if device is not None and device.type == 'cuda':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')
    return dist


def _broadcast_from_rank0(arr):
    """An int64 array decided on rank 0 (shuffled index, epoch seed) becomes every rank's."""
    rank, world = _dist_env()
    if world == 1:
        return arr
    dist = _dist_init()
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(dev)
    dist.broadcast(t, 0)
    return t.cpu().numpy()


def load_data_pretrain(datasets, mode, root='Data/output_pretrain'):
    """pretrain.py:548-576: <root>/<ds>/<ds>_{train,test,valid}_split.npy -> one pool, shuffled (global np.random, unseeded like
    the reference; under torchrun rank 0's draw is broadcast), 85 % / 15 % split. The files stay memory-mapped: the two returned
    objects are index views (data.OctupleShards) with the array protocol the reference's callers use (len, shape, [i])."""
    if mode != "pretrain":
        return None
    arrays = []
    for dataset in datasets:
        parts = OctupleShards.from_files([os.path.join(root, dataset, dataset + '_%s_split.npy' % s) for s in ('train', 'test', 'valid')])
        print(f'   {dataset}: {parts.shape}')
        arrays += parts.arrays
    pool = OctupleShards(arrays)
    print('   > all training data:', pool.shape)
    index = np.arange(len(pool))
    np.random.shuffle(index)
    index = _broadcast_from_rank0(index)
    split = int(len(pool) * 0.85)
    return pool.subset(index[:split]), pool.subset(index[split:])


def _loader_kw(num_workers):
    """Worker processes survive the epochs: forking and importing five of them costs seconds, every epoch, in the reference's
    DataLoader(num_workers=5). No pin_memory: a batch is 0.5 MB of int16, and pinning one per step (hipHostMalloc / hipHostFree)
    synchronises the device -- measured 63.1 -> 65.5 ms per batch."""
    kw = dict(num_workers=num_workers)
    if num_workers > 0:
        kw.update(persistent_workers=True, prefetch_factor=4)
    return kw


def make_loaders(X_train, X_val, batch_size, num_workers, seed=None, balance=True):
    """main.py:28-35. One process: the reference's two DataLoaders. Under torchrun `batch_size` stays the GLOBAL batch (what
    nn.DataParallel scatters, pretrain.py:63-65): each rank loads batch_size / world samples per step through a
    DistributedSampler whose per-epoch permutation is seeded identically on every rank (seed drawn on rank 0)."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    rank, world = _dist_env()
    if world == 1:
        kw = _loader_kw(num_workers)
        return (DataLoader(MidiDataset(X=X_train), batch_size=batch_size, shuffle=True, **kw),
                DataLoader(MidiDataset(X=X_val), batch_size=batch_size, **kw))
    if batch_size % world:
        raise PBError('--batch_size %d is the global batch and must be a multiple of the %d ranks' % (batch_size, world))
    if seed is None:
        seed = int(_broadcast_from_rank0(np.array([np.random.randint(0, 2 ** 31 - 1)]))[0])
    loaders = []
    for X, shuffle in ((X_train, True), (X_val, False)):
        ds = MidiDataset(X=X)
        if balance:
            # the same global batches, dealt to the ranks by sequence length: the packed step's time follows the kept rows, and the
            # ranks meet at every bucket exchange (data.BalancedDistributedSampler)
            sampler = BalancedDistributedSampler(sequence_lengths(ds.data), world, rank, batch_size, shuffle=shuffle, seed=seed)
        else:
            sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed)
        loaders.append(DataLoader(ds, batch_size=batch_size // world, sampler=sampler, **_loader_kw(num_workers)))
    return tuple(loaders)


class Pretrainer:
    """pretrain.py:51-546. Same constructor arguments and public methods."""

    def __init__(self, pianobart: PianoBart, train_dataloader, valid_dataloader, lr, batch, max_seq_len, mask_percent, cpu,
                 cuda_devices=None):
        if cpu or not torch.cuda.is_available():
            raise PBError('pianobart_amd has no CPU execution path: run on an MI355X (got cpu=%s, cuda available=%s)'
                          % (cpu, torch.cuda.is_available()))
        if cuda_devices is not None and len(cuda_devices) > 1:
            raise PBError('nn.DataParallel is replaced by one process per GPU: launch with '
                          '`python -m torch.distributed.run --nproc-per-node N ...` (see INTEGRATION.md)')
        dev_id = cuda_devices[0] if cuda_devices else 0
        if 'LOCAL_RANK' in os.environ:
            dev_id = int(os.environ['LOCAL_RANK'])
        self.device = torch.device('cuda', dev_id)
        torch.cuda.set_device(self.device)
        self.pianobart = pianobart.to(self.device)          # saved alone in the checkpoint (pretrain.py:100)
        self.model = PianoBartLM(pianobart).to(self.device)
        self.total_params = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        print('# total parameters:', self.total_params)
        print("Use GPU", self.device)
        self.engine = self.model._get_engine()
        self.engine.bind(self.device)
        self.engine.pipeline_updates = True                     # the AdamW pass of step i runs beside the forward of step i + 1 (Engine.optimizer_step)
        self.train_data = train_dataloader
        self.valid_data = valid_dataloader
        self.lr = lr
        self.batch = batch
        self.max_seq_len = max_seq_len
        self.mask_percent = mask_percent
        self.quiet = False
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.reducer = None
        self.rank = int(os.environ.get('RANK', 0))
        if self.world > 1:
            from .parallel import GradReducer
            _dist_init(self.device)
            self.reducer = GradReducer(self.engine, self.world)
        self._w = np.array([len(pianobart.e2w[k]) for k in pianobart.e2w], dtype=np.float64)   # e2w dict order (pretrain.py:185-189)
        self._id_limits = np.array(pianobart.n_tokens, dtype=np.int64)                      # table sizes, classes order (PianoBart.py:29-31)
        # corruption stream keyed by the rank: ranks must not draw the same positions (the dropout stream is keyed in Engine)
        self._step_seed = (0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * self.rank) & 0xFFFFFFFFFFFFFFFF
        self._epoch = 0

    # ---- reference API ------------------------------------------------------------------------------
    def train(self):
        self.model.train()
        sampler = getattr(self.train_data, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):                     # DistributedSampler: a new, rank-consistent permutation every epoch
            sampler.set_epoch(self._epoch)
        self._epoch += 1
        return self.iteration(self.train_data, self.max_seq_len)

    def valid(self):
        self.model.eval()
        return self.iteration(self.valid_data, self.max_seq_len, train=False)

    def save_checkpoint(self, epoch, best_acc, valid_acc, valid_loss, train_loss, is_best, filename):
        """pretrain.py:96-110: same dict keys; 'state_dict' holds PianoBart only; optimizer = HF-AdamW state, moments keyed by parameter name."""
        eng = self.engine
        opt = dict(eng.optimizer_state(self.model), lr=self.lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)   # waits for a pipelined update
        # The moments are keyed by PianoBartLM's parameter names ('pianobart.*', 'mask_lm.*') because the optimizer owns the LM heads too,
        # while 'state_dict' below is PianoBart's alone (the reference's file format, pretrain.py:98-106, which drops the heads). So that the
        # moments of the heads have their weights beside them -- and `--resume` continues the SAME trajectory -- the heads travel inside
        # this dict (the top-level keys stay the reference's).
        opt['mask_lm'] = {k: v.detach().cpu() for k, v in self.model.mask_lm.state_dict().items()}
        state = {'epoch': epoch + 1, 'state_dict': {k: v.detach().cpu() for k, v in self.pianobart.state_dict().items()},
                 'best_acc': best_acc, 'valid_acc': valid_acc, 'valid_loss': valid_loss, 'train_loss': train_loss, 'optimizer': opt}
        torch.save(state, filename)
        best_mdl = filename.split('.')[0] + '_best.ckpt'
        if is_best:
            shutil.copyfile(filename, best_mdl)

    def resume(self, filename):
        """Continue from a checkpoint `save_checkpoint` wrote: PianoBart weights ('state_dict'), the LM heads and the AdamW moments / step count
        ('optimizer'). Returns (epoch, best_acc). A reference-written file (torch's optimizer.state_dict(): 'state' / 'param_groups') carries
        neither heads nor named moments and is refused -- load its 'state_dict' as a pre-trained model instead."""
        from .model import checkpoint_state_dict
        ck = torch.load(filename, map_location='cpu', weights_only=False)
        opt = ck.get('optimizer') or {}
        if 'mask_lm' not in opt or not isinstance(opt.get('exp_avg'), (dict, type(None))):
            raise PBError('%s has no resumable optimizer state (written by the reference or before round 6): load its state_dict as a pre-trained model' % filename)
        self.engine.finish_updates()
        self.pianobart.load_state_dict(checkpoint_state_dict(ck['state_dict']), strict=True)
        self.model.mask_lm.load_state_dict(opt['mask_lm'], strict=True)
        self.engine.bind(self.device)
        self.engine.refresh_shadow(force=True)
        self.engine.load_optimizer_state(self.model, opt)
        self._epoch = int(ck.get('epoch', 0))
        return int(ck.get('epoch', 0)), ck.get('best_acc', 0)

    def gen_mask(self, input_ids, choice=None):
        """pretrain.py:211-546 for ONE sequence (S,8): returns (masked (S,8) long, mask (S,) long) on the input's device."""
        ids = input_ids.to(self.device).long().reshape(1, -1, 8)
        enc16, lm, _ = self._corrupt(ops.ids_to_i16(ids), None if choice is None else [choice])
        return enc16[0].long().to(input_ids.device), lm[0, :, 0].long().to(input_ids.device)

    # ---- the step ---------------------------------------------------------------------------------------
    def _corrupt(self, ids16, choices=None):
        B, S = ids16.shape[:2]
        pb = self.pianobart
        if choices is None:
            choices = [random.randint(1, 5) for _ in range(B)]        # the reference's dispatcher draw (pretrain.py:520)
        # the per-sample choices go up through a small pinned ring: torch.tensor(list, device=...) is a pageable, blocking copy, and the
        # batch pipeline stages batch i + 1 on a side stream precisely so as not to wait for the device
        ring = getattr(self, '_choice_pins', None)
        if ring is None or ring[0][0].numel() < B:
            ring = self._choice_pins = [[torch.empty(max(B, 64), dtype=torch.int32).pin_memory() for _ in range(4)], 0]
        ring[1] = (ring[1] + 1) % len(ring[0])
        pin = ring[0][ring[1]][:B]
        pin.copy_(torch.as_tensor(choices, dtype=torch.int32))
        ch = torch.empty(B, dtype=torch.int32, device=self.device)
        ch.copy_(pin, non_blocking=True)
        out = torch.empty_like(ids16)
        lm = torch.empty(B, S, 8, dtype=torch.float32, device=self.device)
        self._step_seed = (self._step_seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        ops.corrupt(ids16, out, lm, ch, None, float(self.mask_percent), self._step_seed, pb.pad_word_np, pb.mask_word_np, pb.n_tokens)
        return out, lm, ch

    def prepare_batch(self, ori_seq_batch):
        """pretrain.py:125-153 on the device: corrupted encoder ids, shift-right decoder ids, loss mask, attention masks."""
        if ori_seq_batch.device.type == 'cpu':                             # nn.Embedding's IndexError (PianoBart.py:15-16), on the loader's host copy
            a = ori_seq_batch.numpy()
            if a.size and (a.min() < 0 or (a.reshape(-1, 8).max(0) >= self._id_limits).any()):
                raise IndexError('index out of range in self: an Octuple id lies outside its embedding table (sizes %s)' % list(self._id_limits))
        ori = ori_seq_batch.to(self.device, non_blocking=True)
        tgt16 = ori.contiguous() if ori.dtype == torch.int16 else ops.ids_to_i16(ori.long() if ori.dtype != torch.int64 else ori)
        if ori_seq_batch.device.type != 'cpu':                             # a batch that is already on the device: checked there, the verdict is read
            tgt16 = self.engine.note_ids(tgt16, owned=tgt16.data_ptr() != ori_seq_batch.data_ptr())   # (a caller's int16 tensor is checked on a copy) without draining the stream (Engine._raise_if_bad_ids); the corrupted and the
            self.engine._queue_id_verdict()                                # shifted ids derive from these and from in-range specials / random tokens
        B, S = tgt16.shape[:2]
        enc16, loss_mask, _ = self._corrupt(tgt16)
        dec16 = torch.empty_like(tgt16)
        ops.shift_right(tgt16, self.engine.sos16, dec16, B, S)
        pad = int(self.pianobart.bar_pad_word)
        emask = (enc16[:, :, 0] != pad).float()
        dmask = (dec16[:, :, 0] != pad).float()
        return enc16, dec16, tgt16, loss_mask, emask, dmask

    def iteration(self, training_data, max_seq_len, train=True):
        eng = self.engine
        total_acc, total_losses, nb = np.zeros(8), 0.0, 0
        def report(sums):
            """The log lines of one step (pretrain.py:198-207) from its 24 sums: the one device -> host read of the step."""
            nonlocal total_acc, total_losses, nb
            s = sums.double().numpy()
            if (s[8:16] == 0).any() or not np.isfinite(s[0:8]).all():
                # pretrain.py:117 divides by the head's mask count: a head without a loss position (or a non-finite loss) makes the loss and
                # every gradient NaN, here as in the reference -- and AdamW then carries the NaN in its moments for good. Say so, loudly.
                sys.stderr.write('[pianobart_amd] WARNING: loss-mask counts %s, loss sums %s: this step poisons the parameters with NaN exactly as the '
                                 'reference does (pretrain.py:116-117); the run cannot recover from it\n' % (s[8:16].tolist(), s[0:8].tolist()))
            losses = s[0:8] / s[8:16]
            accs = s[16:24] / s[8:16]
            total_loss = float((losses * self._w).sum() / self._w.sum())
            if not self.quiet:
                sys.stdout.write('Loss: {:06f} | loss: {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}\n'.format(total_loss, *losses))
                sys.stdout.write('Acc: {:06f} | acc: {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}\n'.format(np.average(accs), *accs))
            total_acc += accs
            total_losses += total_loss
            nb += 1

        # Software pipeline over the batches. Batch i + 1 is corrupted, shifted and masked on a side stream while step i runs, and its
        # packing counts are requested there too (Engine.prefetch_pack): step i + 1 then starts without draining the device. The 24 sums
        # of step i go to a pinned buffer by a stream-ordered copy behind an event; they are read (and the log lines written) after
        # step i + 1 has been handed to the GPU, so neither the read nor the batch preparation sits between two device steps.
        main = torch.cuda.current_stream(self.device)
        if getattr(self, '_prep_stream', None) is None:
            self._prep_stream = torch.cuda.Stream(device=self.device)
            self._sum_pins = [torch.empty(24, dtype=torch.float32).pin_memory() for _ in range(3)]
        prep = self._prep_stream

        def stage(batch):
            prep.wait_stream(main)                                      # everything the previous batches were given (their buffers may be recycled)
            with torch.cuda.stream(prep):
                tensors = self.prepare_batch(batch)
                eng.prefetch_pack(tensors[3], tensors[4], tensors[5], stream=prep)
                ev = torch.cuda.Event()
                ev.record(prep)
            for t in tensors:
                t.record_stream(main)
            return tensors, ev

        def enqueue_read(sums, k):
            """24 floats device -> pinned host behind an event on the launch stream; nothing waits here."""
            pin = self._sum_pins[k % len(self._sum_pins)]
            pin.copy_(sums, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
            return pin, ev

        eng._pack_prefetch.clear()                                      # requests of an epoch that ended early (an exception between stage and step)
        it = iter(training_data)
        first = next(it, None)
        staged = stage(first) if first is not None else None
        pending, k = None, 0
        while staged is not None:
            (enc16, dec16, tgt16, loss_mask, emask, dmask), ready = staged
            main.wait_event(ready)
            nxt = next(it, None)
            staged = stage(nxt) if nxt is not None else None            # on the side stream, beside the step below
            sums = eng.loss_and_grads(enc16, dec16, tgt16, loss_mask, emask, dmask, train=train, ids_checked=True,      # prepare_batch validated the host batch
                                      count_hook=self.reducer.reduce_counts if self.reducer else None)
            if train:
                if self.reducer:
                    self.reducer.all_reduce_grads()
                eng.optimizer_step(lr=self.lr)
            if self.reducer:
                sums = sums.clone()                                     # the engine reuses its scalar buffer in the next step
                self.reducer.reduce_sums(sums)
            cur = enqueue_read(sums, k)
            k += 1
            if pending is not None:
                pending[1].synchronize()
                report(pending[0])
            pending = cur
        if pending is not None:
            pending[1].synchronize()
            report(pending[0])
        n = max(1, len(training_data))
        return round(total_losses / n, 3), [round(float(x) / n, 3) for x in total_acc]


def _load_vocab(dict_file):
    if dict_file.endswith('.json'):
        import json
        e2w = json.load(open(dict_file))['e2w']
        return e2w, {k: {v: w for w, v in d.items()} for k, d in e2w.items()}
    import pickle
    with open(dict_file, 'rb') as f:
        return pickle.load(f)


class _RunLog:
    """result/pretrain/<name>/{model.ckpt, model_best.ckpt, log}: written by rank 0 only; line formats of main.py:84-99."""

    def __init__(self, name, rank0):
        self.dir = 'result/pretrain/' + name
        self.filename = os.path.join(self.dir, 'model.ckpt')
        self.rank0 = rank0
        os.makedirs(self.dir, exist_ok=True)

    def write(self, text):
        if self.rank0:
            with open(os.path.join(self.dir, 'log'), 'a') as outfile:
                outfile.write(text)


def pretrain(argv=None):
    """The pre-training driver (reference: main.py:17-100): same flags, prints, checkpoint and log artefacts; the loop itself is
    organised around the device step (one process per GPU, rank 0 owns the artefacts)."""
    args = get_args_pretrain(argv)
    rank, world = _dist_env()
    if world > 1:                                              # join the job before anything is drawn: rank 0's shuffle is everyone's
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
        torch.cuda.set_device(dev)
        _dist_init(dev)
    print("Loading Dictionary")
    e2w, w2e = _load_vocab(args.dict_file)
    print("\nLoading Dataset", args.datasets)
    X_train, X_val = load_data_pretrain(datasets=args.datasets, mode="pretrain", root=args.data_root)
    train_loader, valid_loader = make_loaders(X_train, X_val, args.batch_size, args.num_workers)
    print("   len of train_loader", len(train_loader))
    print("   len of valid_loader", len(valid_loader))
    print("\nBuilding BART model")
    shape = dict(max_position_embeddings=args.max_seq_len, d_model=args.hs)
    for side in ('encoder', 'decoder'):                        # main.py:39-47: one size for both stacks
        shape.update({side + '_layers': args.layers, side + '_ffn_dim': args.ffn_dims, side + '_attention_heads': args.heads})
    pianobart = PianoBart(bartConfig=BartConfig(**shape), e2w=e2w, w2e=w2e, precision=args.precision)
    print("\nCreating BART Trainer")
    trainer = Pretrainer(pianobart, train_loader, valid_loader, args.lr, args.batch_size, args.max_seq_len, args.mask_percent,
                         args.cpu, args.cuda_devices)
    trainer.quiet = args.quiet
    print("\nTraining Start")
    run = _RunLog(args.name, rank == 0)
    print("   save model at {}".format(run.filename))
    best_acc, stale, first_epoch = 0, 0, 0
    if args.resume:
        first_epoch, best_acc = trainer.resume(args.resume)
        print("   resumed from {} at epoch {} (best_acc {})".format(args.resume, first_epoch, best_acc))
        log_path = os.path.join(run.dir, 'log')
        if os.path.exists(log_path) and not open(log_path).read().endswith('\n'):
            run.write('\n')                                    # the closing line of the earlier run has no newline (main.py:97-99)
    start_t = time.time()
    for epoch in range(first_epoch, args.epochs):
        if stale >= 30:
            print('valid acc not improving for 30 epochs')
            break
        train_loss, train_acc = trainer.train()
        valid_loss, valid_acc = trainer.valid()
        avg_acc = float(np.dot(valid_acc, pianobart.n_tokens)) / sum(pianobart.n_tokens)       # n_tokens-weighted accuracy (main.py:72-74)
        is_best = avg_acc > best_acc
        best_acc = max(avg_acc, best_acc)
        stale = 0 if is_best else stale + 1
        print('epoch: {}/{} | Train Loss: {} | Train acc: {} | Valid Loss: {} | Valid acc: {}'.format(
            epoch + 1, args.epochs, train_loss, train_acc, valid_loss, valid_acc))
        if run.rank0:
            trainer.save_checkpoint(epoch, best_acc, valid_acc, valid_loss, train_loss, is_best, run.filename)
        run.write('Epoch {}: train_loss={}, train_acc={}, valid_loss={}, valid_acc={}\n'.format(epoch + 1, train_loss, train_acc, valid_loss, valid_acc))
    end_t = time.time()
    closing = f'Time cost in pretrain of PianoBart is {end_t - start_t}, start_t = {start_t}, end_t = {end_t}'
    print(closing)
    run.write(closing)
    return trainer


if __name__ == '__main__':
    pretrain()
