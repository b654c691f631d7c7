#!/usr/bin/env python
"""Timing of reduced BASELINE.json configs[3] / [4] shapes (GPT-2 d 256 x 6 layers, L 50, CLM, sampled softmax;
BERT d 512 x 4 layers, L 100, MLM, tied full softmax) on one GPU -- parity-test configurations, measured
for the record / to spot pathological kernels (bench.py reports configs[1])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers4rec_amd as tr
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
if which == "gpt2":
    V, D, NL, NH, L, B = 200_000, 256, 6, 8, 50, 512
    schema = tr.session_schema(V, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="clm", embedding_dim_default=D)
    cfg = tr.GPT2Config.build(D, NH, NL, total_seq_length=L)
    task = tr.NextItemPredictionTask(weight_tying=True, sampled_softmax=True, max_n_samples=100)
else:
    V, D, NL, NH, L, B = 100_000, 512, 4, 8, 100, 256
    schema = tr.session_schema(V, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    cfg = tr.BertConfig.build(D, NH, NL, total_seq_length=L)
    task = tr.NextItemPredictionTask(weight_tying=True)
torch.manual_seed(0)
model = cfg.to_torch_model(inputs, task).to(dev)
dense, tables = tr.flatten_model(model)
opt = tr.FusedAdam([dense, tables], lr=1e-3)
model.train()
batches = [tr.random_data_from_schema(schema, B, L, seed=i, device=dev) for i in range(4)]
def step(i):
    out = model(batches[i % 4], training=True); out["loss"].backward(); opt.step(); return out
for i in range(10): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 30
for i in range(n): out = step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{which}: {B * n / dt:.1f} sessions/s, {1e3 * dt / n:.3f} ms/step, loss {float(out['loss'].detach()):.4f}")
