#!/usr/bin/env python
"""Which Markov generator makes bench.py's Recall@20 / NDCG@20 probe informative (neither at the generator's ceiling, as round
4's single-successor chain, nor on the initial loss plateau)?  Trains the benchmarked configuration for K steps per
(fanout, steps) pair on the HIP path and prints the metrics next to the chain's Bayes ceilings."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda", 0)
for fanout, steps in ((1, 200), (2, 200), (2, 400), (3, 400), (4, 400), (4, 800), (8, 800)):
    bench.MARKOV_FANOUT = fanout
    gen = lambda n, seq, active, seed: bench.markov_sessions(n, seq, active, seed, fanout=fanout)
    tr, schema, model, dense, tables, opt = bench.build(dev, 0.3, lr=2e-3)
    active = 1 + torch.arange(2000) * (bench.V_ITEMS // 2000)
    model.train()
    t0 = time.perf_counter()
    for i in range(steps):
        out = model({"item_id": gen(bench.BATCH, bench.SEQ, active, 10 + i).to(dev)}, training=True)
        out["loss"].backward()
        opt.step()
    model.eval()
    task = model.prediction_task
    task.reset_metrics()
    with torch.no_grad():
        for j in range(4):
            h = model.heads[0].body({"item_id": gen(bench.BATCH, bench.SEQ, active, 900_000 + j).to(dev)}, training=False, testing=True)
            task.evaluate_ranks(h)
    mt = task.compute_metrics()
    rec, ndcg = bench.markov_bayes(fanout=fanout)
    print(json.dumps({"fanout": fanout, "steps": steps, "recall_at_20": round(mt["next-item/recall_at_20"], 4),
                      "ndcg_at_20": round(mt["next-item/ndcg_at_20"], 4), "loss": round(float(out["loss"].detach()), 4),
                      "bayes_recall": round(rec, 4), "bayes_ndcg": round(ndcg, 4), "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    del model, opt, dense, tables
