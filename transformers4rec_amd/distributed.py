"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" ==
RCCL on ROCm, over xGMI).  Replaces the bucketed NCCL all-reduce torch DDP performs for the
reference under HF Trainer (transformers4rec/torch/trainer.py:131-161).

Semantics preserved: each rank's loss is the mean over ITS label rows; gradients
are summed over ranks and divided by world_size (the division is folded into the fused Adam
step).  Two flat buckets: the embedding tables (large; reduced first, on a side stream, while
the transformer backward is still running -- the head is first in backward order) and the dense
rest (small, latency bound).  The same code runs on CPU tensors with the gloo backend (tests).
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, dense_grad, tables_grad=None, group=None):
        self.dense, self.tables, self.group = dense_grad, tables_grad, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._stream = None
        self._pending = None

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def reduce_tables_async(self):
        """Launch the table-bucket all-reduce as soon as the head/input gradients for it exist."""
        if self.world == 1 or self.tables is None:
            return
        if self.tables.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._pending = dist.all_reduce(self.tables, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._pending = dist.all_reduce(self.tables, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_all(self, tables_already_launched=False):
        if self.world == 1:
            return
        if self.tables is not None and not tables_already_launched:
            dist.all_reduce(self.tables, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(self.dense, op=dist.ReduceOp.SUM, group=self.group)
        if self._pending is not None:
            self._pending.wait()
            if self._stream is not None:
                torch.cuda.current_stream().wait_stream(self._stream)
            self._pending = None


def shard_batch(global_batch, rank, world):
    """contiguous B_loc = global_batch / world rows per rank (the reference shards by loader
    partition, transformers4rec/torch/utils/data_utils.py:322-360)."""
    if global_batch % world:
        raise ValueError("global batch must divide by the world size")
    per = global_batch // world
    return rank * per, (rank + 1) * per
