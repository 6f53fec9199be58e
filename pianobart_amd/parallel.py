"""Data-parallel pre-training over RCCL/xGMI: one process per GPU (torch.distributed backend "nccl" is
RCCL on ROCm). Replaces the reference's single-process nn.DataParallel (pretrain.py:63-65), which
broadcasts 966 MB of parameters, gathers 168 MB of logits and reduces 812 MB of gradients through
GPU 0 every step. Here parameters/optimizer are replicated, logits never leave the
rank, and the only exchanges are
  * one 8-float all-reduce of the per-head loss-mask counts (so the loss is the reference's GLOBAL
    sum(ce*mask)/sum(mask), pretrain.py:117, not a mean of per-rank means), and
  * the gradient all-reduce (SUM), issued per layer bucket (~28-38 MB f32) as soon as that layer's
    backward has been enqueued, so RCCL runs on its own stream underneath the rest of backward.
Clip-norm and AdamW then run identically on every rank on the reduced flat buffer.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, engine, world_size, group=None):
        self.eng, self.world, self.group = engine, world_size, group
        self.pending = []
        engine.grad_hook = self._on_ready

    def reduce_counts(self, counts):
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)

    def _on_ready(self, lo, hi):
        """Engine callback: flat gradient range [lo, hi) is final (all producing kernels enqueued)."""
        if hi > lo:
            self.pending.append(dist.all_reduce(self.eng.G32[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def all_reduce_grads(self):
        """Wait (stream-side) for every bucket issued during backward."""
        for w in self.pending:
            w.wait()
        self.pending = []

    def reduce_sums(self, sums):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
