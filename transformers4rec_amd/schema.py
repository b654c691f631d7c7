"""Minimal column schema for the hot path.

The reference builds its input block from a `merlin_standard_lib.Schema`
(transformers4rec/torch/features/sequence.py:140-229, merlin_standard_lib/schema/schema.py).
The schema layer itself is out of scope; this is the small surface
`TabularSequenceFeatures.from_schema` needs: tag selection, names, `int_domain.max`
(cardinality = max + 1, merlin_standard_lib/schema/schema.py:541-550), `value_count.max`.
Any object with the same duck-typed attributes (e.g. the real merlin Schema) is accepted by
`from_schema`.
"""
import enum
import random
from dataclasses import dataclass, field
from typing import List, Optional

import torch


class Tags(enum.Enum):
    CATEGORICAL = "categorical"
    CONTINUOUS = "continuous"
    LIST = "list"
    ITEM_ID = "item_id"
    ITEM = "item"
    TARGET = "target"
    EMBEDDING = "embedding"


@dataclass
class IntDomain:
    min: int = 0
    max: int = 0
    is_categorical: bool = True


@dataclass
class ValueCount:
    min: int = 0
    max: int = 0


@dataclass
class ColumnSchema:
    name: str
    tags: List[Tags] = field(default_factory=list)
    int_domain: Optional[IntDomain] = None
    value_count: Optional[ValueCount] = None


def _tag_values(tags):
    return {getattr(t, "value", t) for t in tags}


class Schema:
    def __init__(self, feature=None):
        self.feature = list(feature or [])

    @classmethod
    def from_json(cls, path_or_text):
        """The reference's schema file (transformers4rec/data/testing/schema.json: the tensorflow-metadata JSON
        form that merlin_standard_lib.Schema.from_json reads, merlin_standard_lib/schema/schema.py): per feature
        `name`, `annotation.tag`, `intDomain {min, max, isCategorical}`, `valueCount {min, max}`.  Accepts a path
        or the JSON text.  Tags this package does not model stay plain strings (tag selection compares values)."""
        import json
        import os

        if isinstance(path_or_text, (str, os.PathLike)) and os.path.exists(str(path_or_text)):
            with open(path_or_text) as f:
                js = json.load(f)
        else:
            js = json.loads(path_or_text)
        known = {t.value: t for t in Tags}
        cols = []
        for ft in js.get("feature", []):
            tags = [known.get(t, t) for t in ft.get("annotation", {}).get("tag", [])]
            dom, vc = ft.get("intDomain"), ft.get("valueCount")
            cols.append(ColumnSchema(
                ft["name"], tags,
                IntDomain(int(dom.get("min", 0)), int(dom["max"]), bool(dom.get("isCategorical", False))) if dom else None,
                ValueCount(int(vc.get("min", 0)), int(vc["max"])) if vc else None))
        return cls(cols)

    @property
    def column_names(self):
        return [c.name for c in self.feature]

    def select_by_tag(self, tags):
        if not isinstance(tags, (list, tuple, set)):
            tags = [tags]
        want = _tag_values(tags)
        return Schema([c for c in self.feature if want & _tag_values(c.tags)])

    def select_by_name(self, names):
        names = [names] if isinstance(names, str) else list(names)
        return Schema([c for c in self.feature if c.name in names])

    def remove_by_name(self, names):
        names = [names] if isinstance(names, str) else list(names)
        return Schema([c for c in self.feature if c.name not in names])

    @property
    def item_id_column_name(self):
        cols = self.select_by_tag(Tags.ITEM_ID).column_names
        if not cols:
            raise ValueError("schema has no item-id column")
        return cols[0]

    def __len__(self):
        return len(self.feature)

    def __iter__(self):
        return iter(self.feature)

    def __add__(self, other):
        return Schema(self.feature + list(other.feature))


def categorical_cardinalities(schema):
    """name -> table rows (int_domain.max + 1)"""
    out = {}
    for col in schema.feature:
        dom = getattr(col, "int_domain", None)
        if dom is not None and getattr(dom, "max", None) is not None and "continuous" not in _tag_values(col.tags):
            out[col.name] = int(dom.max) + 1
    return out


def session_schema(item_cardinality, max_len, categoricals=(), continuous=()):
    """item_cardinality = largest item id (table rows = item_cardinality + 1)."""
    cols = [ColumnSchema("item_id", [Tags.CATEGORICAL, Tags.ITEM_ID, Tags.LIST, Tags.ITEM],
                         IntDomain(0, item_cardinality), ValueCount(1, max_len))]
    for name, card in categoricals:
        cols.append(ColumnSchema(name, [Tags.CATEGORICAL, Tags.LIST], IntDomain(0, card), ValueCount(1, max_len)))
    for name in continuous:
        cols.append(ColumnSchema(name, [Tags.CONTINUOUS, Tags.LIST], None, ValueCount(1, max_len)))
    return Schema(cols)


def random_data_from_schema(schema, num_rows, max_session_length, min_session_length=5, seed=0,
                            device="cpu", ragged=False):
    """Synthetic Schema-driven session batch, following the recipe of the reference's
    transformers4rec/torch/utils/schema_utils.py:29-145: per row len ~ randint(min, max);
    categorical ids ~ randint(1, cardinality); continuous ~ rand; right-padded with 0 to
    max_session_length.  ragged=True returns {name__values, name__offsets} instead."""
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    lens = torch.tensor([rnd.randint(min_session_length, max_session_length) for _ in range(num_rows)])
    m = torch.arange(max_session_length)[None] < lens[:, None]
    cards = categorical_cardinalities(schema)
    out = {}
    for col in schema.feature:
        if col.name in cards:
            x = torch.randint(1, cards[col.name], (num_rows, max_session_length), generator=g) * m
        else:
            x = torch.rand((num_rows, max_session_length), generator=g) * m
        if ragged:
            out[col.name + "__values"] = x[m].to(device)
            out[col.name + "__offsets"] = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(device)
        else:
            out[col.name] = x.to(device)
    return out
