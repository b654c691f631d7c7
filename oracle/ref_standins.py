"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-in modules that let the *unmodified* reference package under
/root/reference (NVIDIA-Merlin/Transformers4Rec, pure Python) import in this
container, where `merlin.*`, `merlin_standard_lib`'s betterproto dependency and
`torchmetrics` are not installed and HuggingFace `transformers` is 5.x (the
reference pins <4.31).

Used by oracle/make_golden.py (fixture generation) and oracle/cpu_reference_bench.py.
/root/reference does not exist on the GPU box; the committed fixtures under tests/golden/ are
what travels -- the reference itself never does, in any form (rounds 3-4 staged a scratch copy for one
GPU call; that script is gone since round 6).

What is faked: only plumbing -- registries, docstring
decorators, schema containers, a minimal torchmetrics.Metric.  No arithmetic of the
hot path is replaced: embeddings, masking, aggregation, HF XLNet/GPT-2/BERT blocks
and the prediction head run the reference's / HF's own code.
"""
import enum
import inspect
import re
import sys
import types

import os


def _reference_root():
    """T4R_REFERENCE_ROOT (a maintainer's own checkout), else /root/reference"""
    return os.environ.get("T4R_REFERENCE_ROOT") or "/root/reference"


REFERENCE_ROOT = _reference_root()


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class Tags(enum.Enum):
    CATEGORICAL = "categorical"
    CONTINUOUS = "continuous"
    LIST = "list"
    SEQUENCE = "sequence"
    ITEM_ID = "item_id"
    ITEM = "item"
    USER_ID = "user_id"
    USER = "user"
    SESSION_ID = "session_id"
    SESSION = "session"
    CONTEXT = "context"
    TEXT = "text"
    TEXT_TOKENIZED = "text_tokenized"
    EMBEDDING = "embedding"
    TARGET = "target"
    BINARY = "binary"
    BINARY_CLASSIFICATION = "binary_classification"
    CLASSIFICATION = "classification"
    MULTI_CLASS_CLASSIFICATION = "multi_class"
    REGRESSION = "regression"
    TIME = "time"
    ID = "id"


def camelcase_to_snakecase(name):
    s1 = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub("([a-z0-9])([A-Z])", r"\1_\2", s1).lower()


class Registry:
    """dict-backed registry with the call surface the reference uses."""

    def __init__(self, registry_name, default_key_fn=None, validator=None, on_set=None,
                 value_transformer=None):
        self._name = registry_name
        self._registry = {}
        self._instantiate = False

    @classmethod
    def class_registry(cls, registry_name, **kw):
        # merlin semantics: registered classes resolve to no-arg instances (made lazily
        # here, because the decorator runs before the class name is bound)
        r = cls(registry_name, **kw)
        r._instantiate = True
        return r

    def __setitem__(self, key, value):
        self._registry[key] = value

    def _resolve(self, key):
        v = self._registry[key]
        if self._instantiate and inspect.isclass(v):
            v = v()
            self._registry[key] = v
        return v

    def register(self, key_or_value=None):
        def decorator(value, key):
            self[key] = value
            return value

        if callable(key_or_value):
            return decorator(key_or_value, camelcase_to_snakecase(key_or_value.__name__))
        return lambda value: decorator(value, key_or_value)

    def register_with_multiple_names(self, *names):
        def decorator(value):
            for n in names:
                self[n] = value
            return value

        return decorator

    def __getitem__(self, key):
        return self._resolve(key)

    def __contains__(self, key):
        return key in self._registry

    def keys(self):
        return self._registry.keys()

    def values(self):
        return self._registry.values()

    def items(self):
        return self._registry.items()

    def parse(self, class_or_str):
        if isinstance(class_or_str, str):
            if class_or_str not in self._registry:
                raise KeyError(f"{class_or_str} never registered with registry {self._name}")
            return self._resolve(class_or_str)
        return class_or_str


def docstring_parameter(*args, **kwargs):
    def dec(obj):
        return obj

    return dec


def filter_kwargs(kwargs, thing_with_kwargs, cascade_kwargs_if_possible=False,
                  argspec_fn=inspect.getfullargspec):
    arg_spec = argspec_fn(thing_with_kwargs)
    if cascade_kwargs_if_possible and arg_spec.varkw is not None:
        return kwargs
    fn_args = list(arg_spec.args)
    return {k: v for k, v in kwargs.items() if k in fn_args}


def has_field(obj, name):
    return getattr(obj, name, None) is not None


class _IntDomain:
    def __init__(self, min=0, max=0, is_categorical=True):
        self.min, self.max, self.is_categorical = min, max, is_categorical


class _ValueCount:
    def __init__(self, min=0, max=0):
        self.min, self.max = min, max


class ColumnSchema:
    def __init__(self, name="", tags=(), int_domain=None, float_domain=None, value_count=None,
                 shape=None, **kw):
        self.name = name
        self.tags = list(tags)
        self.int_domain = int_domain
        self.float_domain = float_domain
        self.value_count = value_count
        self.shape = shape
        self.annotation = types.SimpleNamespace(tag=[getattr(t, "value", t) for t in self.tags])


class Schema:
    """Minimal merlin_standard_lib.Schema: tag / name selection over ColumnSchema."""

    def __init__(self, feature=None):
        self.feature = list(feature or [])

    # import-time calls from transformers4rec.data.* (datasets instantiated at import)
    def from_json(self, *a, **k):
        return self

    def from_proto_text(self, *a, **k):
        return self

    @property
    def column_names(self):
        return [f.name for f in self.feature]

    @property
    def column_schemas(self):
        return self.feature

    def select_by_tag(self, tags):
        if not isinstance(tags, (list, tuple, set)):
            tags = [tags]
        out = [f for f in self.feature if any(t in f.tags for t in tags)]
        return Schema(out)

    def remove_by_tag(self, tags):
        if not isinstance(tags, (list, tuple, set)):
            tags = [tags]
        return Schema([f for f in self.feature if not any(t in f.tags for t in tags)])

    def select_by_name(self, names):
        if isinstance(names, str):
            names = [names]
        return Schema([f for f in self.feature if f.name in names])

    def remove_by_name(self, names):
        if isinstance(names, str):
            names = [names]
        return Schema([f for f in self.feature if f.name not in names])

    def filter_columns_from_dict(self, d):
        return {k: v for k, v in d.items() if k in self.column_names}

    @property
    def item_id_column_name(self):
        cols = self.select_by_tag(Tags.ITEM_ID).column_names
        if not cols:
            raise ValueError("no item-id column")
        return cols[0]

    def categorical_cardinalities(self):
        return categorical_cardinalities(self)

    def __add__(self, other):
        return Schema(self.feature + other.feature)

    def __iter__(self):
        return iter(self.feature)

    def __len__(self):
        return len(self.feature)

    def __bool__(self):
        return True

    def copy(self, **kw):
        return Schema(list(self.feature))


def categorical_cardinalities(schema):
    # reference: merlin_standard_lib/schema/schema.py:541-550  (int_domain.max + 1)
    out = {}
    for col in schema.feature:
        if col.int_domain is not None:
            out[col.name] = col.int_domain.max + 1
    return out


def get_embedding_sizes_from_schema(schema, multiplier=2.0):
    cards = categorical_cardinalities(schema)
    return {k: int(max(16, multiplier * (v ** 0.25))) for k, v in cards.items()}


def install():
    """Register every stand-in in sys.modules and put the reference on sys.path."""
    import torch

    if "transformers4rec" in sys.modules:
        return
    # ---- merlin.*
    _mod("merlin")
    ms = _mod("merlin.schema")
    ms.Tags = Tags
    ms.TagsType = object
    ms.Schema = Schema
    ms.ColumnSchema = ColumnSchema
    ms.TagSet = set
    mst = _mod("merlin.schema.tags")
    mst.Tags = Tags
    mst.TagsType = object
    _mod("merlin.schema.io")
    pu = _mod("merlin.schema.io.proto_utils")
    pu.has_field = has_field
    pu.copy_better_proto_message = lambda *a, **k: None
    tfm = _mod("merlin.schema.io.tensorflow_metadata")
    tfm.TensorflowMetadata = type("TensorflowMetadata", (), {})
    _mod("merlin.models")
    _mod("merlin.models.utils")
    du = _mod("merlin.models.utils.doc_utils")
    du.docstring_parameter = docstring_parameter
    rg = _mod("merlin.models.utils.registry")
    rg.Registry = Registry
    rg.camelcase_to_snakecase = camelcase_to_snakecase
    mu = _mod("merlin.models.utils.misc_utils")
    mu.filter_kwargs = filter_kwargs
    mu.validate_dataset = lambda *a, **k: None
    _mod("merlin.models.utils.schema_utils")
    _mod("merlin.dataloader")
    dl = _mod("merlin.dataloader.torch")
    dl.Loader = type("Loader", (), {})
    # ---- merlin_standard_lib
    msl = _mod("merlin_standard_lib")
    msl.Schema = Schema
    msl.ColumnSchema = ColumnSchema
    msl.categorical_cardinalities = categorical_cardinalities
    _mod("merlin_standard_lib.schema")
    mss = _mod("merlin_standard_lib.schema.schema")
    mss.Schema = Schema
    mss.ColumnSchema = ColumnSchema
    mss.categorical_cardinalities = categorical_cardinalities
    _mod("merlin_standard_lib.utils")
    eu = _mod("merlin_standard_lib.utils.embedding_utils")
    eu.get_embedding_sizes_from_schema = get_embedding_sizes_from_schema
    _mod("merlin_standard_lib.proto")
    sbp = _mod("merlin_standard_lib.proto.schema_bp")
    for n in ("Feature", "FeatureType", "FixedShape", "FloatDomain", "IntDomain", "ValueCount",
              "ValueCountList", "Annotation"):
        setattr(sbp, n, type(n, (), {}))
    msl.schema = sys.modules["merlin_standard_lib.schema"]
    msl.utils = sys.modules["merlin_standard_lib.utils"]

    # ---- torchmetrics
    tm = _mod("torchmetrics")

    class Metric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default
            setattr(self, name, [] if isinstance(default, list) else default.clone())

        def reset(self):
            for n, d in self._defaults.items():
                setattr(self, n, [] if isinstance(d, list) else d.clone())

        def forward(self, *a, **k):
            return self.update(*a, **k)

    tm.Metric = Metric
    for n in ("Precision", "Recall", "Accuracy", "MeanSquaredError"):
        setattr(tm, n, lambda *a, **k: Metric())
    _mod("torchmetrics.regression").MeanSquaredError = tm.MeanSquaredError
    _mod("torchmetrics.utilities")
    tud = _mod("torchmetrics.utilities.data")
    tud.dim_zero_cat = lambda x: torch.cat(x, dim=0) if isinstance(x, (list, tuple)) else x

    # ---- transformers 5.x shims for symbols removed since the reference's <4.31 pin
    import transformers
    import transformers.modeling_utils as tmu

    if not hasattr(transformers, "TFTrainingArguments"):
        transformers.TFTrainingArguments = type("TFTrainingArguments", (), {})
    if not hasattr(transformers, "TransfoXLConfig"):
        transformers.TransfoXLConfig = type(
            "TransfoXLConfig", (transformers.PretrainedConfig,), {"model_type": "transfo-xl"}
        )
    if not hasattr(tmu, "SequenceSummary"):

        class SequenceSummary(torch.nn.Module):
            def __init__(self, config):
                super().__init__()
                self.summary_type = getattr(config, "summary_type", "last")

            def forward(self, hidden_states, cls_index=None):
                if self.summary_type == "last":
                    return hidden_states[:, -1]
                if self.summary_type == "first":
                    return hidden_states[:, 0]
                return hidden_states.mean(dim=1)

        tmu.SequenceSummary = SequenceSummary

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    """Returns the reference's `transformers4rec.torch` module (unmodified source)."""
    install()
    import transformers4rec.torch as tr  # noqa: E402

    return tr
