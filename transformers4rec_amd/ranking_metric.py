"""Ranking metrics of the next-item task as functions of the TARGET'S RANK.

Mirror of transformers4rec/torch/ranking_metric.py:30-280 for the case the task uses them in
(`labels_onehot=True`, prediction_task.py:338-343): every label row has exactly ONE relevant item, so
with r = the 0-based rank of that item among the V scores (ties to the lower index, as torch.topk
breaks them) every metric of the reference collapses to a function of r:

    PrecisionAt     precision@k = [r < k] / k                  (:73-98)
    RecallAt        recall@k    = [r < k]                      (:107-147; one relevant item)
    AvgPrecisionAt  ap@k        = [r < k] / (r + 1)            (:151-192: the only non-zero term of
                                                                 sum_j precision@j * rel_j is j = r + 1,
                                                                 and num_relevant.clamp(1, k) = 1)
    DCGAt           dcg@k       = [r < k] / log2(r + 2)        (:196-239)
    NDCGAt          ndcg@k      = dcg@k / 1                    (:242-280: the ideal DCG of one item is 1)

so neither the [N, V] one-hot (:52-59) nor a top-k list is needed -- the rank comes out of the logits
GEMM's epilogue (ops.rank_of_target) or from a fused top-k.  The reference classes are torchmetrics
modules whose state is `cat`-synchronised over the ranks and averaged over all rows (:50, :64-66);
here the state is a (sum, count) pair per metric and cut-off, all-reduced in
NextItemPredictionTask.compute_metrics -- the same mean.
"""
import re

import torch


def _snake(name):
    """merlin.models.utils.registry.camelcase_to_snakecase, the naming rule of PredictionTask.metric_name
    (torch/model/base.py:221-222): NDCGAt -> ndcg_at, AvgPrecisionAt -> avg_precision_at"""
    s1 = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub("([a-z0-9])([A-Z])", r"\1_\2", s1).lower()


class RankingMetric:
    """Descriptor of one metric family at several cut-offs; `from_ranks(ranks)` -> float32 [N, len(top_ks)]."""

    def __init__(self, top_ks=None, labels_onehot=False):
        top_ks = top_ks or [2, 5]
        if not isinstance(top_ks, (list, tuple)):
            top_ks = [top_ks]
        self.top_ks = [int(k) for k in top_ks]
        self.labels_onehot = labels_onehot

    @property
    def name(self):
        return _snake(type(self).__name__)

    def value(self, ranks_f, hit, k):
        raise NotImplementedError

    def from_ranks(self, ranks):
        r = ranks.to(torch.float32)
        return torch.stack([self.value(r, ranks < k, k) for k in self.top_ks], dim=1)

    def __repr__(self):
        return f"{type(self).__name__}(top_ks={self.top_ks})"


class PrecisionAt(RankingMetric):
    def value(self, r, hit, k):
        return hit.to(torch.float32) / float(k)


class RecallAt(RankingMetric):
    def value(self, r, hit, k):
        return hit.to(torch.float32)


class AvgPrecisionAt(RankingMetric):
    def value(self, r, hit, k):
        return torch.where(hit, 1.0 / (r + 1.0), torch.zeros_like(r))


class DCGAt(RankingMetric):
    def value(self, r, hit, k):
        return torch.where(hit, 1.0 / torch.log2(r + 2.0), torch.zeros_like(r))


class NDCGAt(DCGAt):
    pass


_BY_NAME = {c.__name__: c for c in (PrecisionAt, RecallAt, AvgPrecisionAt, DCGAt, NDCGAt)}
# registry names of the reference (ranking_metric.py:72, 106, 150, 195, 241)
_ALIASES = {"precision_at": PrecisionAt, "precision": PrecisionAt, "recall_at": RecallAt, "recall": RecallAt,
            "avg_precision_at": AvgPrecisionAt, "avg_precision": AvgPrecisionAt, "map": AvgPrecisionAt,
            "dcg_at": DCGAt, "dcg": DCGAt, "ndcg_at": NDCGAt, "ndcg": NDCGAt}


def default_metrics(top_ks=(10, 20)):
    """NextItemPredictionTask.DEFAULT_METRICS (prediction_task.py:338-343), in the reference's order"""
    ks = list(top_ks)
    return (NDCGAt(top_ks=ks, labels_onehot=True), AvgPrecisionAt(top_ks=ks, labels_onehot=True),
            RecallAt(top_ks=ks, labels_onehot=True))


def coerce(metric, top_ks=(10, 20)):
    """A metric of this module, a registry name of the reference, or one of the REFERENCE's metric objects
    (recognised by class name; its `top_ks` are kept) -> a RankingMetric of this module.  Anything that is not
    a function of the target's rank is off the path and raises."""
    if isinstance(metric, RankingMetric):
        return metric
    if isinstance(metric, str):
        if metric not in _ALIASES:
            raise NotImplementedError(f"ranking metric {metric!r} is not on the HIP path (supported: {sorted(_ALIASES)})")
        return _ALIASES[metric](top_ks=list(top_ks), labels_onehot=True)
    cls = _BY_NAME.get(type(metric).__name__)
    if cls is None or not getattr(metric, "top_ks", None):
        raise NotImplementedError(f"metric {type(metric).__name__} is not a rank-based metric of the HIP path "
                                  f"(supported: {sorted(_BY_NAME)})")
    return cls(top_ks=list(metric.top_ks), labels_onehot=getattr(metric, "labels_onehot", True))
