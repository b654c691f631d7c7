"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" ==
RCCL on ROCm, over xGMI).  Replaces the bucketed NCCL all-reduce torch DDP performs for the
reference under HF Trainer (transformers4rec/torch/trainer.py:131-161).

Semantics preserved: each rank's loss is the mean over ITS label rows; gradients
are summed over ranks and divided by world_size (the division is folded into the fused Adam
step).  Three kinds of traffic:

  dense bucket    everything but the tables (3.4 MB at C2): one all-reduce, latency bound.
  tables bucket   the embedding tables / untied output layer as ONE dense buffer: all-reduce.  With a
                  tied full-softmax head the head's d W fills every row, and that part is final right
                  after the head's backward -- `reduce_tables_async()` launches the all-reduce there
                  (async: on the process group's stream), under the transformer's backward (the head is FIRST
                  in backward order).  Stream budget: caller + two weight-gradient streams + the collective's = 4;
                  a fifth active stream slows the whole step down (profiles/r05_q_stream_count.txt).
  row-sparse      gradients that touch few rows -- the lookup scatter of the input block (B*L rows),
                  the sampled-softmax head (labels + negatives) -- are never scattered into the dense
                  bucket while the all-reduce may be in flight: a SparseRowExchange collects them as
                  (ids, rows), all-gathers both, and every rank applies ALL ranks' rows to its
                  (already reduced) table gradient with the deterministic sorted scatter
                  (csrc/embedding_sorted.hip), in rank order -- bit-identical on every rank, so the
                  replicas cannot drift.  With a sampled / untied head nothing dense is left in the
                  tables bucket: pass tables_grad=None and 1 GB (C4) of all-reduce disappears.

The same code runs on CPU tensors with the gloo backend (tests; `apply_fn` is injected there since
the HIP scatter needs a GPU).
"""
import torch
import torch.distributed as dist


def _hip_apply(d_table, ids, rows, padding_idx):
    from . import ops

    ops.scatter_rows_sorted(d_table, ids, rows, padding_idx)


class SparseRowExchange:
    """Sink for row-sparse table gradients.  `attach(param)` routes the HIP backward of that table
    (features._table_scatter, the sampled head) here instead of into `param.grad`; `exchange()` --
    called by GradReducer.reduce_all after the dense reductions -- makes every rank apply every rank's
    rows.  grad_of(param) -> the buffer to add into (default: param.grad, created if missing)."""

    def __init__(self, group=None, apply_fn=None, grad_of=None, equal_sizes=False):
        # equal_sizes: every rank contributes the same number of rows to every exchange (the lookup scatter of equal
        # per-rank batches: B * L rows) -- the per-step size all-gather and its host synchronisation are skipped.
        # Leave False whenever the counts can differ (a sampled head's label rows).
        self.equal_sizes = equal_sizes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.apply_fn = apply_fn or _hip_apply
        self.grad_of = grad_of
        self._pending = []
        self.bytes_exchanged = 0

    def attach(self, *params):
        for p in params:
            p._t4r_sparse_sink = self
        return self

    @staticmethod
    def detach(*params):
        for p in params:
            if hasattr(p, "_t4r_sparse_sink"):
                del p._t4r_sparse_sink

    # ---- called from the backward pass
    def add(self, tab, ids, grad_rows, col, dim, ids_div, padding_idx):
        """lookup scatter of one feature: grad_rows [n * ids_div, W]; the rows of this feature are the
        columns [col, col + dim), summed over the ids_div positions a per-session id was broadcast to"""
        from . import ops

        W = grad_rows.shape[-1]
        n = ids.numel()
        if ids_div > 1:
            rows = ops.seq_sum_cols(grad_rows, col, dim, n, ids_div)
        elif W != dim:
            rows = ops.copy_cols_out(grad_rows, col, dim)
        else:
            rows = grad_rows
        self.add_rows(tab, ids.reshape(-1), rows, padding_idx)

    def add_rows(self, tab, ids, rows, padding_idx=-1):
        self._pending.append((tab, ids.contiguous(), rows.contiguous(), int(padding_idx)))

    # ---- called once per step, after backward
    def _target(self, tab):
        if self.grad_of is not None:
            return self.grad_of(tab)
        if tab.grad is None:
            tab.grad = torch.zeros_like(tab)
        return tab.grad

    def exchange(self):
        pending, self._pending = self._pending, []
        for tab, ids, rows, pad in pending:
            tgt = self._target(tab)
            if self.world == 1:
                self.apply_fn(tgt, ids, rows, pad)
                continue
            if self.equal_sizes:
                nmax = ids.numel()
            else:
                n = torch.tensor([ids.numel()], device=ids.device, dtype=torch.int64)
                sizes = [torch.zeros_like(n) for _ in range(self.world)]
                dist.all_gather(sizes, n, group=self.group)
                nmax = max(int(s.item()) for s in sizes)
            D = rows.shape[1]
            if ids.numel() < nmax:      # ranks with fewer rows (label counts differ) pad with ignored ids
                fill = pad if pad >= 0 else tgt.shape[0]           # out-of-range id: carries no gradient
                ids = torch.cat([ids, torch.full((nmax - ids.numel(),), fill, device=ids.device, dtype=ids.dtype)])
                rows = torch.cat([rows, torch.zeros((nmax - rows.shape[0], D), device=rows.device, dtype=rows.dtype)])
            ids_all = torch.empty(self.world * nmax, device=ids.device, dtype=ids.dtype)
            rows_all = torch.empty((self.world * nmax, D), device=rows.device, dtype=rows.dtype)
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(ids_all, ids, group=self.group)
                dist.all_gather_into_tensor(rows_all, rows, group=self.group)
            else:       # gloo (tests): the list form, straight into the rank-major slices
                dist.all_gather(list(ids_all.chunk(self.world)), ids, group=self.group)
                dist.all_gather(list(rows_all.chunk(self.world)), rows, group=self.group)
            self.bytes_exchanged += ids_all.numel() * 8 + rows_all.numel() * 4
            # rank-major concatenation + stable sort => summation order (rank, lookup), the same everywhere
            self.apply_fn(tgt, ids_all, rows_all, pad)


def collective_channels():
    """CUs an RCCL kernel of this process may hold while it runs: one workgroup per channel.  RCCL sizes its channel count from
    the topology unless NCCL_MAX_NCHANNELS caps it; `bench.py` sets the cap (default 16) before the process group exists, so
    that the number is known here.  Unknown (no cap in the environment): 32 is assumed."""
    import os

    try:
        return max(1, int(os.environ.get("NCCL_MAX_NCHANNELS", "32")))
    except ValueError:
        return 32


class GradReducer:
    def __init__(self, dense_grad, tables_grad=None, group=None, sparse: SparseRowExchange = None):
        self.dense, self.tables, self.group, self.sparse = dense_grad, tables_grad, group, sparse
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending = None
        self._launched = False
        self._budgeted = False

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def reduce_tables_async(self):
        """Launch the table-bucket all-reduce as soon as its dense part is final (right after the head's
        backward).  Only legal when everything that arrives later goes through `sparse` -- otherwise the
        late writers would race with the reduction."""
        if self.world == 1 or self.tables is None:
            return
        self._launched = True
        if self.tables.is_cuda:
            # async_op=True: the process group runs the collective on ITS OWN stream, ordered after everything enqueued on the
            # current stream so far -- no stream of ours in between.  That matters: the step already drives the caller's
            # stream and the two weight-gradient streams of the body's backward, the collective's stream is the fourth, and a
            # FIFTH active stream slows every kernel of the step down (+1.0 ms per step measured on one GPU with one tiny
            # launch per step on an extra stream; GPU_MAX_HW_QUEUES does not lift it: profiles/r05_q_stream_count.txt).
            # Everything that writes the tables bucket before this point (the head's d W) runs on the caller's stream: c10d's
            # ordering is the only dependency.  (the opt-in side stream of the head's d W, which this function used
            # to drain into the caller's stream, is gone -- prediction_task.py.)
            self._pending = dist.all_reduce(self.tables, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            # the ring kernel now holds one CU per channel until the reduction is done, i.e. for most of the body's backward:
            # its one-workgroup-per-CU token-tile kernels are told to plan for the CUs that are left (csrc/xlnet_fused.hip:
            # t4r_xlnet_set_cu_budget; measured on one GPU with a CU occupier, tools/occupier_curve.py: 1.36x -> 1.25x, DESIGN.md
            # section 6).  Only an RCCL collective holds CUs: gloo / CPU groups set no budget.  Cleared by reduce_all (in a
            # finally), by the end-of-backward callback queued here, and -- if the backward raised before either -- by the next
            # training forward (transformer.XLNetModel.forward: ops.xlnet_clear_cu_budget).
            if dist.get_backend(self.group) == "nccl":
                from . import ops

                ops.xlnet_set_cu_budget(max(1, ops.device_cus() - collective_channels()))
                self._budgeted = True
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(self._clear_budget)
                except RuntimeError:
                    pass                    # not inside a backward pass: reduce_all clears it
        else:
            self._pending = dist.all_reduce(self.tables, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _clear_budget(self):
        if self._budgeted:          # the backward pass is over: the forward's kernels get the whole chip again
            from . import ops

            ops.xlnet_set_cu_budget(0)
            self._budgeted = False

    def reduce_all(self, tables_already_launched=None):
        launched = self._launched if tables_already_launched is None else tables_already_launched
        self._launched = False
        try:
            if self.world > 1:
                if self.tables is not None and not launched:
                    dist.all_reduce(self.tables, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(self.dense, op=dist.ReduceOp.SUM, group=self.group)
                if self._pending is not None:
                    self._pending.wait()
                    self._pending = None
        finally:
            self._clear_budget()
        if self.sparse is not None:
            self.sparse.exchange()      # world 1: the local deterministic scatter


def head_backward_hook(model, fn):
    """Calls fn() when the gradient w.r.t. the prediction head's input exists, i.e. right after the head's
    backward and before the transformer body's (the point where a tied table's dense gradient is final).
    Returns the hook handle (a forward hook on the transformer block that registers a tensor hook)."""
    def fwd_hook(mod, inputs, output):
        if torch.is_tensor(output) and output.requires_grad:
            output.register_hook(lambda g: (fn(), None)[1])

    return model.transformer_block.register_forward_hook(fwd_hook)


def shard_batch(global_batch, rank, world):
    """contiguous B_loc = global_batch / world rows per rank (the reference shards by loader
    partition, transformers4rec/torch/utils/data_utils.py:322-360)."""
    if global_batch % world:
        raise ValueError("global batch must divide by the world size")
    per = global_batch // world
    return rank * per, (rank + 1) * per
