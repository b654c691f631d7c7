#!/usr/bin/env python
"""Timing of BASELINE.json configs[2] (C3 = C2 + 3 categoricals + 2 continuous soft embeddings, concat,
ReLU projection to d_model) on one GPU -- a parity-test configuration, measured for the record only
(bench.py reports configs[1])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers4rec_amd as tr
V, D, NL, NH, L, B = 100_000, 128, 4, 4, 20, 1024
dev = torch.device("cuda", 0)
cats, conts = (("category", 1000), ("brand", 100), ("dow", 10)), ("price", "age")
schema = tr.session_schema(V, L, cats, conts)
torch.manual_seed(0)
inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", aggregation="concat",
                                                d_output=D, continuous_soft_embeddings=True,
                                                embedding_dims={"item_id": D}, embedding_dim_default=64)
cfg = tr.XLNetConfig.build(D, NH, NL, total_seq_length=L, dropout=0.3)
model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True)).to(dev)
dense, tables = tr.flatten_model(model)
opt = tr.FusedAdam([dense, tables], lr=1e-3)
model.train()
batches = [tr.random_data_from_schema(schema, B, L, seed=i, device=dev) for i in range(8)]
def step(i):
    out = model(batches[i % 8], training=True); out["loss"].backward(); opt.step(); return out
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0: step(0)
for i in range(20): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 100
for i in range(n): out = step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"C3 (336-wide concat -> 128): {B * n / dt:.1f} sessions/s, {1e3 * dt / n:.3f} ms/step, loss {float(out['loss'].detach()):.4f}")
