"""Data-parallel engine: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" IS RCCL on ROCm; "gloo" for the CPU tests).

The reference has no distributed code at all; semantics here
are "same as one process with the global batch": every rank holds a full
replica, takes a contiguous shard of the global batch, and after backward the
flat fp32 gradient buffer (ONE bucket, 73.8 MB for NACF/MSRVTT-shape) is
all-reduced (sum); the 1/world scale is folded into the fused Adam launch, so
clip(+-5) follows the reduce exactly as misc/run.py:258-261 orders them.
Per-rank losses are normalised by the LOCAL batch (misc/crit.py:40), so the
mean of rank gradients equals the global-batch gradient.  BatchNorm statistics
are per-rank (documented deviation; the fusion layer sees 7 680 rows per rank).
"""
import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int):
    """contiguous shard [lo, hi) of the global batch owned by `rank`"""
    assert global_batch % world == 0, 'global batch must divide evenly over ranks'
    per = global_batch // world
    return rank * per, (rank + 1) * per


class DataParallel(object):
    def __init__(self, model, process_group=None):
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def broadcast_parameters(self, src=0):
        if self.world == 1:
            return
        dist.broadcast(self.model.flat.data, src, group=self.group)
        for b in self.model.buffers():
            dist.broadcast(b, src, group=self.group)

    def all_reduce_gradients(self):
        """sum-all-reduce of the single flat gradient bucket"""
        if self.world == 1:
            return
        dist.all_reduce(self.model.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
