"""Device-resident session feed: parquet list columns -> ragged (values, offsets)
in HBM -> padded `[B, L]` batches, assembled by one HIP launch per feature.

Replaces, for the hot path, the reference's MerlinDataLoader + `pad_batch` map
(transformers4rec/torch/utils/data_utils.py:216-494, utils/padding.py:72-122):

  * wire format: parquet with `list<int64>` / `list<float>` columns for sequential features and
    plain int64 / float columns for per-session (context) features
    (transformers4rec/data/testing/data.parquet is of this form);
  * every list feature is right-zero-padded / truncated to `max_sequence_length`
    (pad_batch semantics: fixed length, padding index 0);
  * partitioning for data parallelism: contiguous row ranges per rank (`global_rank` of
    `global_size`), the reference splits at partition level (data_utils.py:322-360);
  * iteration contract read by the reference's training loops (trainer.py:232,453-456):
    `__iter__`, `__len__`, `.dataset` (Sized), `._batch_size`.

The host only parses the file (pyarrow) once; with 288 GB of HBM the whole dataset stays on the
device and a batch costs no host->device traffic: a batch is a vector of row ids (a slice of a
device permutation when shuffling) and `t4r_ragged_gather_to_padded` per feature.
"""
from typing import Dict, Iterable, Optional, Sequence, Union

import numpy as np
import torch

from . import ops


def read_ragged_columns(paths: Union[str, Sequence[str]], columns: Optional[Iterable[str]] = None):
    """parquet -> {name: (values ndarray, offsets int64 ndarray | None)} on the host.
    List columns give (flat values, offsets[rows + 1]); scalar columns give (values[rows], None).
    Null lists count as empty; integers widen to int64, floats narrow to float32 (the dtypes the
    path computes in, model/base.py:546-548)."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    paths = [paths] if isinstance(paths, str) else list(paths)
    tables = [pq.read_table(p, columns=None if columns is None else list(columns)) for p in paths]
    table = pa.concat_tables(tables) if len(tables) > 1 else tables[0]
    out = {}
    for name in table.column_names:
        col = table.column(name).combine_chunks()
        if pa.types.is_list(col.type) or pa.types.is_large_list(col.type):
            if col.null_count:
                col = col.fill_null([])
            offs = np.asarray(col.offsets, dtype=np.int64)
            vals = col.values.to_numpy(zero_copy_only=False)
            vals = vals[offs[0]: offs[-1]]      # a sliced ListArray keeps its parent's values
            offs = offs - offs[0]
        else:
            offs = None
            vals = col.to_numpy(zero_copy_only=False)
        if np.issubdtype(vals.dtype, np.integer):
            vals = vals.astype(np.int64, copy=False)
        elif np.issubdtype(vals.dtype, np.floating):
            vals = vals.astype(np.float32, copy=False)
        else:
            raise TypeError(f"column {name}: unsupported dtype {vals.dtype}")
        out[name] = (np.array(vals, copy=True), offs)    # own, writable buffers (arrow memory is read-only)
    rows = {len(v) if o is None else len(o) - 1 for v, o in out.values()}
    if len(rows) > 1:
        raise ValueError(f"columns disagree on the number of rows: {rows}")
    return out


class _Sized:
    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n


class ParquetSessionLoader:
    """Iterates `{feature: Tensor[B, L] | Tensor[B]}` batches resident on `device`.

    paths: parquet file(s); schema: optional `tr.Schema` (selects the columns); columns: explicit
    column list otherwise.  shuffle: new device permutation per epoch (seed + epoch).
    with_targets: yield `(inputs, None)` tuples, the shape `Model.fit` iterates
    (torch/model/base.py:669-718).  drop_uneven (with global_size > 1): True = equal shards, trailing
    n % world rows dropped with a warning (training); False = every row kept, shard sizes differ by at most
    one (evaluation loaders)."""

    def __init__(self, paths, batch_size: int, max_sequence_length: int, schema=None, columns=None,
                 device="cuda", shuffle=False, drop_last=False, global_size: Optional[int] = None,
                 global_rank: Optional[int] = None, seed: int = 0, with_targets: bool = False,
                 drop_uneven: bool = True):
        if batch_size <= 0 or max_sequence_length <= 0:
            raise ValueError("batch_size and max_sequence_length must be positive")
        if schema is not None and columns is None:
            columns = list(schema.column_names)
        host = read_ragged_columns(paths, columns)
        n = next(len(v) if o is None else len(o) - 1 for v, o in host.values())
        lo, hi = 0, n
        if global_size is not None and global_size > 1:
            if global_rank is None or not 0 <= global_rank < global_size:
                raise ValueError("global_rank must be in [0, global_size)")
            per = n // global_size
            if per == 0:
                raise ValueError(f"{n} rows cannot be sharded over {global_size} ranks")
            if drop_uneven:
                # TRAINING: every rank gets the SAME number of rows (floor(n / world); the n % world trailing
                # rows are dropped): the data-parallel step is one blocking all-reduce per batch, so ranks with
                # different batch counts would deadlock at the end of the epoch
                lo, hi = global_rank * per, (global_rank + 1) * per
                if n % global_size:
                    import warnings

                    warnings.warn(f"ParquetSessionLoader: {n % global_size} trailing row(s) of {n} dropped so that the "
                                  f"{global_size} ranks see equal shards (drop_uneven=False keeps them, for evaluation)")
            else:
                # EVALUATION (no per-batch collective: the metric sums are reduced once, in compute_metrics): balanced
                # contiguous shards covering every row, the first n % world ranks take one row more
                lo = global_rank * per + min(global_rank, n % global_size)
                hi = lo + per + (1 if global_rank < n % global_size else 0)
        self.device = torch.device(device)
        self._batch_size = batch_size
        self.batch_size = batch_size
        self.max_sequence_length = max_sequence_length
        self.shuffle, self.drop_last, self.seed = shuffle, drop_last, seed
        self.with_targets = with_targets
        self._epoch = 0
        self._row0, self._rows = lo, hi - lo
        self.dataset = _Sized(self._rows)
        self._cols: Dict[str, tuple] = {}
        for name, (vals, offs) in host.items():
            v = torch.from_numpy(vals).to(self.device)
            o = None if offs is None else torch.from_numpy(offs).to(self.device)
            self._cols[name] = (v, o)

    def __len__(self):
        if self.drop_last:
            return self._rows // self._batch_size
        return (self._rows + self._batch_size - 1) // self._batch_size

    def set_epoch(self, epoch: int):
        self._epoch = epoch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator(device=self.device)
            g.manual_seed(self.seed + self._epoch)
            order = torch.randperm(self._rows, generator=g, device=self.device) + self._row0
        else:
            order = torch.arange(self._row0, self._row0 + self._rows, device=self.device)
        self._epoch += 1
        for b in range(len(self)):
            ids = order[b * self._batch_size: (b + 1) * self._batch_size]
            batch = {name: ops.ragged_gather_to_padded(v, o, ids, self.max_sequence_length)
                     for name, (v, o) in self._cols.items()}
            yield (batch, None) if self.with_targets else batch
