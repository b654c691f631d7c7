"""transformers4rec_amd -- MI355X (gfx950) native session-sequence hot path of Transformers4Rec:
TabularSequenceFeatures -> TransformerBlock (XLNet) -> NextItemPredictionTask.

Hand-written HIP kernels behind a C ABI (include/t4r_hip.h, lib/libt4r_hip.so); this package is
the host-side mirror of the reference's module interface.  No CPU fallback exists.
"""
__version__ = "0.1.0"
