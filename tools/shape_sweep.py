"""Shape sweep of the module mirror on one GPU: one training step (forward, backward, Adam) and one inference call per shape,
for shapes the reference accepts (any total_seq_length / d_model / n_head: config/transformer.py:218-260, :432-482, :493-534).
Prints one line per shape: ok / the exception.  Not a test: a survey of where the HIP path stops.

    python tools/shape_sweep.py
"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformers4rec_amd as tr  # noqa: E402

DEV = torch.device("cuda", 0)
CASES = [
    ("xlnet", "mlm", 20, 64, 4), ("xlnet", "mlm", 20, 96, 4), ("xlnet", "clm", 20, 96, 2), ("xlnet", "mlm", 50, 256, 8),
    ("xlnet", "mlm", 64, 128, 4), ("xlnet", "mlm", 65, 128, 4), ("xlnet", "clm", 128, 64, 2), ("xlnet", "mlm", 200, 32, 1),
    ("xlnet", "mlm", 254, 32, 2), ("xlnet", "clm", 255, 32, 2), ("xlnet", "mlm", 20, 320, 2), ("xlnet", "mlm", 20, 448, 8),
    ("xlnet", "mlm", 20, 100, 4), ("xlnet", "mlm", 20, 512, 2),
    ("gpt2", "clm", 20, 64, 4), ("gpt2", "clm", 129, 64, 2), ("gpt2", "clm", 50, 192, 4), ("gpt2", "clm", 300, 32, 2),
    ("gpt2", "clm", 20, 320, 2), ("gpt2", "clm", 600, 32, 2), ("xlnet", "mlm", 500, 32, 2), ("bert", "mlm", 1023, 16, 1),
    ("bert", "mlm", 20, 64, 4), ("bert", "mlm", 200, 48, 2), ("bert", "mlm", 100, 512, 8), ("bert", "mlm", 20, 96, 2),
    ("xlnet", "mlm", 100, 64, 4, "concat"), ("xlnet", "clm", 300, 32, 2, "concat"), ("xlnet", "mlm", 100, 64, 2, "element-wise-sum"),
    ("gpt2", "clm", 150, 64, 2, "concat"), ("bert", "mlm", 20, 192, 4, "concat"),
]


def run(arch, masking, L, D, H, multi=None, V=500, B=6, dropout=0.1):
    torch.manual_seed(0)
    if multi:           # BASELINE configs[2]'s input block: categoricals + continuous soft embeddings, concat or element-wise sum
        schema = tr.session_schema(V, L, (("category", 40), ("brand", 9)), ("price", "age") if multi == "concat" else ())
        kw = dict(max_sequence_length=L, masking=masking, aggregation=multi)
        if multi == "concat":
            kw.update(continuous_soft_embeddings=True, d_output=D, embedding_dims={"item_id": D}, embedding_dim_default=24)
        else:
            kw.update(embedding_dim_default=D)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    else:
        schema = tr.session_schema(V, L)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking=masking, embedding_dim_default=D)
    if arch == "xlnet":
        cfg = tr.XLNetConfig.build(D, H, 2, total_seq_length=L, dropout=dropout)
    elif arch == "gpt2":
        cfg = tr.GPT2Config.build(D, H, 2, total_seq_length=L)
    else:
        cfg = tr.BertConfig.build(D, H, 1, total_seq_length=L)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV)
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables] if tables is not None else [dense], lr=1e-3)
    x = {k: v.to(DEV) for k, v in tr.random_data_from_schema(schema, B, L, seed=3).items()}
    model.train()
    losses = []
    for _ in range(2):
        out = model(x, training=True)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    model.eval()
    with torch.no_grad():
        scores = model(x)
    assert torch.isfinite(scores).all() and scores.shape[0] == B
    return losses


def main():
    bad = 0
    for case in CASES:
        try:
            losses = run(*case)
            print(f"{case}: ok  loss {losses[0]:.4f} -> {losses[1]:.4f}")
        except Exception as e:          # noqa: BLE001
            bad += 1
            msg = str(e).splitlines()[0][:200] if str(e) else traceback.format_exc().splitlines()[-1]
            print(f"{case}: {type(e).__name__}: {msg}")
    print(f"{len(CASES) - bad} of {len(CASES)} shapes ran")


if __name__ == "__main__":
    main()
