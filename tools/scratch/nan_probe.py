import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import transformers4rec_amd as tr
DEV = "cuda"
torch.manual_seed(0)
B, L, V, D = 1024, 20, 100_000, 128
scale_hi = float(os.environ.get("SCALE_HI", "99"))
lr = float(os.environ.get("LR", "1e-2"))
schema = tr.session_schema(V, L)
inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
cfg = tr.XLNetConfig.build(D, 4, 4, total_seq_length=L, dropout=0.0)
model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
model.to(DEV)
table = model.input_features.item_embedding_table.weight
with torch.no_grad():
    g = torch.Generator(device=DEV).manual_seed(5)
    table.mul_(1.0 + scale_hi * torch.rand(V + 1, 1, device=DEV, generator=g) ** 3)
dense, tables = tr.flatten_model(model)
opt = tr.FusedAdam([dense, tables], lr=lr)
caps = {}
hooks = [model.input_features.register_forward_hook(lambda m, a, o: caps.__setitem__("emb", o.detach())),
         model.transformer_block.register_forward_hook(lambda m, a, o: caps.__setitem__("hid", o.detach()))]
for i, layer in enumerate(model.transformer_block.transformer.layer):
    pass
for step in range(4):
    data = tr.random_data_from_schema(schema, B, L, seed=20 + step)
    out = model({"item_id": data["item_id"].to(DEV)}, training=True)
    print(f"step {step}: loss {float(out['loss']):.5f} emb finite {bool(torch.isfinite(caps['emb']).all())} |emb|max {float(caps['emb'].abs().max()):.3g} "
          f"hid finite {bool(torch.isfinite(caps['hid']).all())} |hid|max {float(caps['hid'].abs().max()):.3g} "
          f"pred finite {bool(torch.isfinite(out['predictions']).all())} |pred|max {float(out['predictions'].abs().max()):.3g}", flush=True)
    out["loss"].backward()
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print("   non-finite grads:", bad[:12], flush=True)
    for n, p in model.named_parameters():
        if p.grad is not None and "layer.0" in n or "embedding_tables" in n:
            print(f"      {n}: |g|max {float(p.grad.abs().max()):.3g}")
    if bad:
        break
    opt.step()
    torch.cuda.synchronize()
    badp = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
    print("   non-finite params after Adam:", badp[:12], flush=True)
    if badp:
        break
