"""Thin caller of the hot path: the reference's Model -> Head -> SequentialBlock composition
(transformers4rec/torch/model/base.py:371-425, 544-598; block/base.py:236-262) reduced to the
single-head next-item case, with the reference's module tree names so state_dicts interchange:
  heads.0.body.0 (TabularSequenceFeatures), heads.0.body.1 (TransformerBlock),
  heads.0.prediction_task_dict.next-item (NextItemPredictionTask).
Rewriting Model/Head/Trainer is out of scope; this exists so the path can
be driven stand-alone (tests, bench) where the reference package is not installed.
"""
from typing import Optional

import torch
from torch import nn

from . import ops


class _Body(nn.ModuleList):
    @property
    def inputs(self):
        return self[0]

    def forward(self, inputs, training=False, testing=False, **kwargs):
        x = self[0](inputs, training=training, testing=testing)
        return self[1](x)


class Head(nn.Module):
    def __init__(self, body: _Body, task):
        super().__init__()
        self.body = body
        self.prediction_task_dict = nn.ModuleDict({task.task_name: task})


class Model(nn.Module):
    def __init__(self, input_features, transformer_block, prediction_task, max_sequence_length: Optional[int] = None,
                 top_k: Optional[int] = None):
        super().__init__()
        body = _Body([input_features, transformer_block])
        hidden = transformer_block.transformer.config.hidden_size
        prediction_task.build(body=body, input_size=(-1, -1, hidden), inputs=input_features)
        self.heads = nn.ModuleList([Head(body, prediction_task)])
        self.max_sequence_length = max_sequence_length or getattr(input_features, "max_sequence_length", None)
        self.top_k = top_k

    @property
    def input_features(self):
        return self.heads[0].body[0]

    @property
    def transformer_block(self):
        return self.heads[0].body[1]

    @property
    def prediction_task(self):
        return next(iter(self.heads[0].prediction_task_dict.values()))

    def pad_inputs(self, inputs):
        """pad_inputs (torch/utils/padding.py:126-164): ragged __values/__offsets -> dense [B, L],
        L = min(max_sequence_length, longest row in the batch)."""
        ragged = [k[: -len("__offsets")] for k in inputs if k.endswith("__offsets")]
        if not ragged:
            return inputs
        L = 0
        for n in ragged:
            L = max(L, int(ops.ragged_max_len(inputs[n + "__offsets"].view(-1).contiguous()).item()))
        if self.max_sequence_length is not None:
            L = min(L, self.max_sequence_length)
        out = {k: v for k, v in inputs.items() if not (k.endswith("__offsets") or k.endswith("__values"))}
        for n in ragged:
            out[n] = ops.ragged_to_padded(inputs[n + "__values"].view(-1).contiguous(),
                                          inputs[n + "__offsets"].view(-1).contiguous(), L)
        return out

    def forward(self, inputs, targets=None, training=False, testing=False, **kwargs):
        inputs = {k: (v.to(torch.float32) if torch.is_floating_point(v) else v) for k, v in inputs.items()}
        inputs = self.pad_inputs(inputs)
        head = self.heads[0]
        h = head.body(inputs, training=training, testing=testing)
        task = self.prediction_task
        if training or testing:
            return task(h, targets=targets, training=training, testing=testing)
        return task(h, training=False, testing=False, top_k=self.top_k)

    def calculate_metrics(self, predictions, targets):
        return self.prediction_task.calculate_metrics(predictions, targets)

    def compute_metrics(self, mode=None):
        return self.prediction_task.compute_metrics(mode)

    def reset_metrics(self):
        self.prediction_task.reset_metrics()
