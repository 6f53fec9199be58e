"""Input side of the pre-train path: the reference stores `(N, 1024, 8)` Octuple arrays as int64 `.npy`
(Data/data_generation/convert.py:560-565), concatenates and shuffles them in memory (pretrain.py:548-576) and wraps them row by
row in `torch.tensor` (dataset.py:4-16). All Octuple ids are < 262, so the same data as int16 is 4x smaller and is exactly what
the kernels read (one 16-byte row per token).

`OctupleShards` keeps every `.npy` file memory-mapped where it lies and addresses sequences through ONE index array: the
reference's concatenate / shuffle / 85-15 split become operations on that index (no copy of the data), and a sample is
materialised as an int16 tensor only when a batch asks for it. `MidiDataset` keeps the reference's name and semantics
(an array or a path), returning int16 rows. `convert_to_int16` rewrites a reference file as an int16 shard (checked lossless)."""
import os

import numpy as np
import torch


def convert_to_int16(src_npy, dst_npy):
    """int64 (or any int) (N,S,8) .npy  ->  int16 .npy (checked lossless)."""
    a = np.load(src_npy, allow_pickle=True)
    b = a.astype(np.int16)
    if not np.array_equal(a, b):
        raise ValueError('%s holds ids outside the int16 range' % src_npy)
    np.save(dst_npy, b)
    return b.shape


def _open(path):
    """Memory-map a plain integer .npy (an `<name>.i16.npy` sibling written by convert_to_int16 is preferred); object arrays (which
    cannot be mapped) are loaded."""
    i16 = path[:-4] + '.i16.npy'
    if os.path.exists(i16):
        path = i16
    try:
        return np.load(path, mmap_mode='r')
    except ValueError:
        return np.asarray(np.load(path, allow_pickle=True).tolist(), dtype=np.int64)


def _row_i16(a, i):
    row = np.asarray(a[i])
    if row.dtype != np.int16:
        r16 = row.astype(np.int16)
        if not np.array_equal(r16, row):
            raise ValueError('Octuple id outside the int16 range')
        row = r16
    return torch.from_numpy(np.array(row, dtype=np.int16))          # a private, writable copy of the 16 KB row (the map is read-only)


class OctupleShards(torch.utils.data.Dataset):
    """Sequences of several (N_k, S, 8) arrays addressed through one index: element j is global row index[j] of the virtual
    concatenation. `subset(idx)` shares the arrays."""

    def __init__(self, arrays, index=None):
        self.arrays = list(arrays)
        self.starts = np.cumsum([0] + [len(a) for a in self.arrays])
        self.index = np.arange(self.starts[-1]) if index is None else np.asarray(index)

    @classmethod
    def from_files(cls, paths):
        return cls([_open(p) for p in paths])

    def subset(self, idx):
        return OctupleShards(self.arrays, self.index[np.asarray(idx)])

    def __len__(self):
        return len(self.index)

    @property
    def shape(self):
        return (len(self),) + tuple(self.arrays[0].shape[1:])

    def __getitem__(self, j):
        g = int(self.index[j])
        k = int(np.searchsorted(self.starts, g, side='right')) - 1
        return _row_i16(self.arrays[k], g - int(self.starts[k]))


class MidiDataset(torch.utils.data.Dataset):
    """dataset.py:4-16. `X` may be an in-memory array, an OctupleShards view, or the path of an .npy shard (opened memory-mapped).
    Items are int16 (S, 8) tensors."""

    def __init__(self, X):
        self.data = _open(X) if isinstance(X, (str, os.PathLike)) else X

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        if isinstance(self.data, OctupleShards):
            return self.data[index]
        return _row_i16(self.data, index)


def sequence_lengths(X, pad_bar=256):
    """Number of non-PAD rows of every sequence of an (N, S, 8) array / OctupleShards view (bar column != PAD, convert.py:331-332)."""
    if isinstance(X, OctupleShards):
        per = [np.asarray((np.asarray(a[:, :, 0]) != pad_bar).sum(axis=1), dtype=np.int64) for a in X.arrays]
        return np.concatenate(per)[X.index] if per else np.zeros(0, dtype=np.int64)
    return np.asarray((np.asarray(X)[:, :, 0] != pad_bar).sum(axis=1), dtype=np.int64)


class BalancedDistributedSampler(torch.utils.data.Sampler):
    """DistributedSampler whose GLOBAL batches are dealt to the ranks by sequence length.

    Same epoch permutation on every rank (seed + epoch) and the same global batches as torch's DistributedSampler would form from it
    (consecutive runs of `global_batch` indices, padded by wrap-around to a multiple of the world size), so the gradient a step sums
    over the replicas is the gradient of the same set of samples (pretrain.py:63-65: nn.DataParallel scatters one global batch). Only
    WHICH rank gets which sample changes: inside a global batch the samples are sorted by their non-PAD length and dealt in snake
    order (0..W-1, W-1..0, ...). The packed step drops a sample's PAD rows, so a rank's step time follows its samples' lengths
    (measured: correlation 0.95, spread 9.9 % over random batches of 32) and the ranks meet at every bucket exchange."""

    def __init__(self, lengths, num_replicas, rank, global_batch, shuffle=True, seed=0):
        if global_batch % num_replicas:
            raise ValueError('global batch %d is not a multiple of the %d ranks' % (global_batch, num_replicas))
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.world, self.rank, self.gb, self.shuffle, self.seed, self.epoch = num_replicas, rank, global_batch, shuffle, seed, 0
        n = len(self.lengths)
        self.total = -(-n // num_replicas) * num_replicas
        self.num_samples = self.total // num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        n = len(self.lengths)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g).numpy()
        else:
            perm = np.arange(n)
        if self.total > n:
            perm = np.concatenate([perm, perm[:self.total - n]])
        mine = []
        for s in range(0, self.total, self.gb):
            batch = perm[s:s + self.gb]
            order = batch[np.argsort(-self.lengths[batch], kind='stable')]
            per = len(order) // self.world
            for j in range(per):                                    # snake deal: round j hands out W consecutive lengths
                k = j * self.world + (self.rank if j % 2 == 0 else self.world - 1 - self.rank)
                mine.append(int(order[k]))
        return iter(mine)
