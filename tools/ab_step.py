"""Same-process, same-box A/B of the training step (BASELINE configs[1]) between two settings of a module attribute:

    python tools/ab_step.py transformer._FUSE_FINAL 1 0 [--steps 200] [--rounds 3]

alternates `rounds` times between the two values (bench.build + make_train_step rebuilt for each run), prints ms per step.
Attributes are looked up under transformers4rec_amd (e.g. transformer._FUSE_FINAL, prediction_task._HEAD_SPLIT)."""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(steps, warm=30):
    dev = torch.device("cuda", 0)
    tr, schema, model, dense, tables, opt = bench.build(dev, 0.3)
    model.train()
    masking = model.input_features.masking
    masking.seed, model.transformer_block.transformer.seed = bench.rank_seeds(0)
    batches = [{k: v.to(dev) for k, v in tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i).items()} for i in range(8)]
    reducer = tr.GradReducer(dense.grad, tables.grad if tables is not None else None)
    step = bench.make_train_step(model, batches, reducer, opt)
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("attr")
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    modname, attr = args.attr.rsplit(".", 1)
    mod = importlib.import_module("transformers4rec_amd." + modname)
    cast = type(getattr(mod, attr))
    conv = (lambda v: v not in ("0", "False", "false")) if cast is bool else cast
    run(50)      # heat
    for r in range(args.rounds):
        for v in (args.a, args.b):
            setattr(mod, attr, conv(v))
            print(f"round {r}: {args.attr} = {v}: {run(args.steps):.4f} ms per step", flush=True)


if __name__ == "__main__":
    main()
