#!/usr/bin/env python
"""BASELINE.json configs[3] / configs[4] AT THEIR NAMED SIZE, one GPU's share (per-GPU batch 1024 of the 8-GPU
global batch 8192) -- parity-test configurations (tests/test_e2e_gpu.py checks them against the oracle);
timed here for the record (bench.py reports configs[1]):

  c4     GPT-2 d_model 256, 6 layers, 4 heads, item vocab 1 M, seq 50, causal LM, sampled softmax (100 negatives),
         tied weights, dropout 0.3, Adam.  The table gradient is row-sparse end to end (lookup scatter + sampled
         head as (ids, rows) through the deterministic sorted scatter).
  c5     BERT d_model 512, 12 layers, 8 heads, item vocab 10 M (20 GB fp32 table), seq 100, MLM, tied FULL softmax
         on the non-materialising head (15 k label rows x 10 M items would be 600 GB of logits), HF dropouts 0.1.
         Precision by argument: fp32 accuracy ("auto") or the reference's AMP mode for this config ("fp16" / "bf16").

    python tools/c45_bench.py c4 [steps]
    python tools/c45_bench.py c5 [steps] [precision] [batch]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import transformers4rec_amd as tr
from transformers4rec_amd import ops

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if which == "c4" else 3)
prec = sys.argv[3] if len(sys.argv) > 3 else "auto"
torch.manual_seed(0)
with torch.device(dev):          # parameters are created in HBM (a 20 GB table takes minutes to initialise on the host)
    if which == "c4":
        V, D, NL, NH, L, B = 1_000_000, 256, 6, 4, 50, 1024
        schema = tr.session_schema(V, L)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="clm", embedding_dim_default=D)
        cfg = tr.GPT2Config.build(D, NH, NL, total_seq_length=L)
        task = tr.NextItemPredictionTask(weight_tying=True, sampled_softmax=True, max_n_samples=100)
    else:
        V, D, NL, NH, L = 10_000_000, 512, 12, 8, 100
        B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
        schema = tr.session_schema(V, L)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
        cfg = tr.BertConfig.build(D, NH, NL, total_seq_length=L)
        task = tr.NextItemPredictionTask(weight_tying=True)      # head_mode "auto" -> fused at this size
    model = cfg.to_torch_model(inputs, task)
ops.set_precision(prec)
dense, tables = tr.flatten_model(model)
opt = tr.FusedAdam([dense, tables], lr=1e-3)
model.train()
batches = [tr.random_data_from_schema(schema, B, L, seed=i, device=dev) for i in range(2)]


def step(i):
    out = model(batches[i % 2], training=True)
    out["loss"].backward()
    opt.step()
    return out


torch.cuda.reset_peak_memory_stats()
out = step(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
n_lab = 0
for i in range(n):
    out = step(1 + i)
    n_lab += out["labels"].numel()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
N = n_lab // n
res = {"config": which, "V": V, "d_model": D, "layers": NL, "seq_len": L, "batch": B, "precision": ops.get_precision(),
       "head_mode": task.resolve_head_mode(N, V + 1) if which == "c5" else "sampled",
       "steps": n, "ms_per_step": round(1e3 * dt / n, 2), "sessions_per_s": round(B * n / dt, 1),
       "label_rows_per_step": N, "loss": round(float(out["loss"].detach()), 4),
       "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
       "logits_if_materialised_GB": round(4.0 * N * (V + 1) / 2 ** 30, 1)}
if which == "c5":
    flops_head = 4 * 2.0 * N * (V + 1) * D          # fwd + recompute + dX + dW
    res["head_TFLOP_per_step"] = round(flops_head / 1e12, 1)
print(json.dumps(res))
