"""Data-parallel pre-training over RCCL/xGMI: one process per GPU (torch.distributed backend "nccl" is
RCCL on ROCm). Replaces the reference's single-process nn.DataParallel (pretrain.py:63-65), which
broadcasts 966 MB of parameters, gathers 168 MB of logits and reduces 812 MB of gradients through
GPU 0 every step. Here parameters/optimizer are replicated, logits never leave the
rank, and the only exchanges are
  * one 8-float all-reduce of the per-head loss-mask counts (so the loss is the reference's GLOBAL
    sum(ce*mask)/sum(mask), pretrain.py:117, not a mean of per-rank means), and
  * the gradient exchange, issued per layer bucket (~28-38 MB of f32 gradients) as soon as that layer's
    backward has been enqueued, on a communication stream of its own, underneath the rest of backward.
Clip-norm and AdamW then run identically on every rank on the reduced flat buffer.

Gradient exchange. Mode "f32" (default since round 6): a plain RCCL all-reduce (SUM) of the f32 bucket -- the arithmetic of the
reference's fp32 reduce_add_coalesced (pretrain.py:65), 812 MB per step at cfg 2. Mode "bf16" (PB_DP_GRADS=bf16; SURVEY 5 budgets 406 MB
per step, half of f32; narrower than the reference, so on request only):
xGMI on an MI355X node is a full mesh of point-to-point links (7 per GPU), so a bucket is exchanged as
a direct reduce-scatter + all-gather in which every link carries 1/N of the bucket in each phase:
  1. the bucket's f32 gradients are rounded to bf16 and cut into N equal chunks;
  2. all-to-all: rank j receives chunk j from every rank (N-1 links busy in both directions);
  3. rank j sums its N received chunks in f32 (pb_sum_rows_bf16) and rounds once to bf16;
  4. all-gather of the reduced chunks; every rank (the owner included) converts the same bf16 values back
     to f32, so all ranks hold bit-identical gradients.
Accumulation is in f32 (one rounding of the inputs, one of the sum), unlike a bf16 ring all-reduce
that rounds after every hop.
"""
import os

import torch
import torch.distributed as dist


class _HipXfer:
    """The elementwise steps of the exchange on the HIP device (C ABI kernels; no other implementation in the product)."""

    @staticmethod
    def to_bf16(src_f32, dst_bf16):
        from . import ops
        ops.cast_f32_to_bf16(src_f32, dst_bf16)

    @staticmethod
    def sum_rows(src_bf16, dst_bf16, rows):
        from . import ops
        ops.sum_rows_bf16(src_bf16, dst_bf16, rows)

    @staticmethod
    def to_f32(src_bf16, dst_f32):
        from . import ops
        ops.cast_bf16_to_f32(src_bf16, dst_f32)


class GradReducer:
    def __init__(self, engine, world_size, group=None, mode=None, xfer=None):
        self.eng, self.world, self.group = engine, world_size, group
        self.mode = mode or os.environ.get('PB_DP_GRADS', 'f32')
        if self.mode not in ('bf16', 'f32'):
            raise ValueError('gradient exchange mode must be "bf16" or "f32"')
        self.xfer = xfer or _HipXfer
        self.pending = []
        self.ranges = []                 # (lo, hi) of every bucket announced since the last all_reduce_grads (tests)
        self._bufs = {}
        self._comm = None
        engine.grad_hook = self._on_ready
        from . import engine as _E
        dev = getattr(engine, 'device', None)
        if _E._DP_RESERVE_CUS > 0 and dev is not None and dev.type == 'cuda':
            from ._lib import LIB
            LIB.call('pb_gemm_reserve_cus', _E._DP_RESERVE_CUS)      # the persistent GEMM grids leave RCCL its CUs (Engine._bwd_dbg)

    def close(self):
        """Uninstall: the engine steps alone again (whole-chip persistent grids)."""
        self.eng.grad_hook = None
        dev = getattr(self.eng, 'device', None)
        if dev is not None and dev.type == 'cuda':
            from ._lib import LIB
            LIB.call('pb_gemm_reserve_cus', 0)

    def reduce_counts(self, counts):
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)

    def _comm_stream(self, dev):
        if self._comm is None and dev.type == 'cuda':
            self._comm = torch.cuda.Stream(device=dev)
        return self._comm

    def _staging(self, key, n, dev):
        b = self._bufs.get(key)
        if b is None or b.numel() < n:
            b = torch.empty(n, dtype=torch.bfloat16, device=dev)
            self._bufs[key] = b
        return b[:n]

    def _on_ready(self, lo, hi):
        """Engine callback: flat gradient range [lo, hi) is final (all producing kernels enqueued on the current stream)."""
        if hi <= lo:
            return
        self.ranges.append((lo, hi))
        g = self.eng.G32[lo:hi]
        if self.mode == 'f32':
            self.pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        W, n, dev = self.world, hi - lo, g.device
        per = (-(-n // W) + 7) // 8 * 8                      # chunk length, a multiple of 8 elements (16-byte vectors)
        k = len(self.ranges)                                 # one set of staging buffers per bucket of a step: buckets overlap in time
        send, recv = self._staging(('s', k), W * per, dev), self._staging(('r', k), W * per, dev)
        own, full = self._staging(('o', k), per, dev), self._staging(('f', k), W * per, dev)
        comm = self._comm_stream(dev)
        if comm is not None:
            ev = torch.cuda.Event()
            ev.record()                                      # the producing stream (main, or the engine's second stream)
            comm.wait_event(ev)
            ctx = torch.cuda.stream(comm)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if W * per > n:
                send[n:].zero_()
            self.xfer.to_bf16(g, send[:n])
            dist.all_to_all_single(recv, send, group=self.group)          # RCCL orders itself after, and the comm stream waits for it
            self.xfer.sum_rows(recv, own, W)
            dist.all_gather_into_tensor(full, own, group=self.group)
            self.xfer.to_f32(full[:n], g)
            if comm is not None:
                done = torch.cuda.Event()
                done.record()
                self.pending.append(done)

    def all_reduce_grads(self):
        """Wait (stream-side) for every bucket issued during backward."""
        for w in self.pending:
            if isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(w)
            else:
                w.wait()
        self.pending = []
        self.ranges = []

    def reduce_sums(self, sums):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
