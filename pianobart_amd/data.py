"""Input side of the pre-train path: the reference stores `(N, 1024, 8)` Octuple arrays as int64 `.npy`
(Data/data_generation/convert.py:560-565) and wraps them row by row in `torch.tensor` (dataset.py:4-16). All Octuple ids are
< 262, so the same data as memory-mapped int16 shards is 4x smaller, needs no parsing and is exactly what the kernels read
(one 16-byte row per token). `MidiDataset` keeps the reference's name and semantics; int16 rows go to the device as they are."""
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


class MidiDataset(torch.utils.data.Dataset):
    """dataset.py:4-16. `X` may be an in-memory array or the path of an .npy shard (opened memory-mapped)."""

    def __init__(self, X):
        self.data = np.load(X, mmap_mode='r') if isinstance(X, (str, os.PathLike)) else X

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return torch.from_numpy(np.ascontiguousarray(self.data[index]))
