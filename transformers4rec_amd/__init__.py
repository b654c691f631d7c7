"""transformers4rec_amd -- MI355X (gfx950) native session-sequence hot path of Transformers4Rec:
TabularSequenceFeatures -> TransformerBlock (XLNet) -> NextItemPredictionTask.

Hand-written HIP kernels behind a C ABI (include/t4r_hip.h, lib/libt4r_hip.so); this package is
the host-side mirror of the reference's module interface.  No CPU fallback exists.
"""
__version__ = "0.1.0"

from .schema import ColumnSchema, IntDomain, Schema, Tags, ValueCount, random_data_from_schema, session_schema  # noqa: E402,F401
from .masking import (CausalLanguageModeling, MaskedLanguageModeling, MaskSequence, GradCarrier,  # noqa: E402,F401
                      enable_autograd_gradients, hip_parameters)
from .features import (  # noqa: E402,F401
    ContinuousFeatures, EmbeddingFeatures, FeatureConfig, SequenceEmbeddingFeatures, SoftEmbedding,
    SoftEmbeddingFeatures, TableConfig, TabularSequenceFeatures)
from . import ranking_metric  # noqa: E402,F401
from .ranking_metric import AvgPrecisionAt, DCGAt, NDCGAt, PrecisionAt, RecallAt  # noqa: E402,F401
from .transformations import StochasticSwapNoise, TabularDropout, TabularLayerNorm  # noqa: E402,F401
from .transformer import TransformerBlock, XLNetConfig, XLNetModel  # noqa: E402,F401
from .transformer_hf import BertConfig, BertModel, GPT2Config, GPT2Model  # noqa: E402,F401
from .prediction_task import LogUniformSampler, NextItemPredictionTask  # noqa: E402,F401
from .model import Head, Model  # noqa: E402,F401
from .optim import FlatParams, FusedAdam, flatten_model  # noqa: E402,F401
from .distributed import GradReducer, SparseRowExchange, head_backward_hook, shard_batch  # noqa: E402,F401
from .data import ParquetSessionLoader, read_ragged_columns  # noqa: E402,F401
from .rng import default_seed, get_rng_state, set_rng_state  # noqa: E402,F401
