"""GPU side of the lockstep comparison at the BENCHMARKED configuration (VERDICT r5 next #1b, #1c).

  (1) `--steps` training steps of BASELINE configs[1] at dropout 0.3 on the HIP path with fixed Philox keys, loss recorded per
      step, Recall@20 / NDCG@20 on the four held-out batches of bench.recall_probe -> `<out>`; oracle/cpu_lockstep.py replays
      the same trajectory on the CPU oracle from the keys alone and merges the two records;
  (2) `--seeds N`: the same procedure from N different (init, MLM key, dropout key) triples -> mean / sd / min / max of the
      final loss, Recall@20 and NDCG@20: the spread any two independent runs of this procedure are expected to show.

    python tools/lockstep_bench_config.py [--steps 200] [--seeds 8] [--out gpurun_out/r06_lockstep_hip.json]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(device, steps, init_seed, mask_seed, drop_seed, dropout, config="c2"):
    tr, schema, model, dense, tables, opt = bench.build(device, dropout, lr=2e-3, config=config, seed=init_seed)
    table = model.input_features.item_embedding_table.weight
    q0 = model.transformer_block.transformer.layer[0].rel_attn.q
    checksum = [round(float(table.detach().double().abs().sum()), 6), round(float(q0.detach().double().abs().sum()), 9)]
    model.input_features.masking.seed = mask_seed
    model.transformer_block.transformer.seed = drop_seed
    active = 1 + torch.arange(2000) * (bench.V_ITEMS // 2000)
    to_dev = lambda d: {k: v.to(device) for k, v in d.items()}
    model.train()
    losses = []
    t0 = time.perf_counter()
    for i in range(steps):
        x = to_dev(bench.session_features(bench.markov_sessions(bench.BATCH, bench.SEQ, active, 10 + i), config))
        out = model(x, training=True)
        out["loss"].backward()
        opt.step()
        losses.append(out["loss"].detach())
    losses = [float(v) for v in torch.stack(losses).cpu()]
    model.eval()
    task = model.prediction_task
    task.reset_metrics()
    with torch.no_grad():
        for j in range(4):
            x = to_dev(bench.session_features(bench.markov_sessions(bench.BATCH, bench.SEQ, active, 900_000 + j), config))
            task.evaluate_ranks(model.heads[0].body(x, training=False, testing=True))
    mt = task.compute_metrics()
    torch.cuda.synchronize()
    return {"init_checksum": checksum, "loss_per_step": [round(v, 6) for v in losses],
            "recall_at_20": round(mt["next-item/recall_at_20"], 4), "ndcg_at_20": round(mt["next-item/ndcg_at_20"], 4),
            "final_train_loss": round(losses[-1], 4), "train_steps": steps, "seconds": round(time.perf_counter() - t0, 2)}


def spread(rows, key):
    v = [r[key] for r in rows]
    return {"mean": round(statistics.mean(v), 4), "sd": round(statistics.stdev(v), 4) if len(v) > 1 else None,
            "min": min(v), "max": max(v), "n": len(v), "values": v}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_lockstep_hip.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    res = run(dev, args.steps, 0, 1234, 4321, args.dropout)
    res.update(what="HIP path, BASELINE configs[1] at dropout %.1f, Adam lr 2e-3, bench.markov_sessions; init seed 0, MLM key 1234, "
                    "dropout key 4321 (the trajectory oracle/cpu_lockstep.py replays)" % args.dropout,
               measured_on=torch.cuda.get_device_name(0))
    # same keys again: the trajectory is bit-reproducible
    again = run(dev, args.steps, 0, 1234, 4321, args.dropout)
    res["bit_reproducible"] = again["loss_per_step"] == res["loss_per_step"] and again["recall_at_20"] == res["recall_at_20"]
    if args.seeds:
        rows = [run(dev, args.steps, 100 + s, 5000 + 17 * s, 9000 + 31 * s, args.dropout) for s in range(args.seeds)]
        res["seed_spread"] = {"what": f"{args.seeds} independent runs of the same procedure (init seed 100+s, MLM key 5000+17s, dropout key "
                                      f"9000+31s), {args.steps} steps each",
                              "recall_at_20": spread(rows, "recall_at_20"), "ndcg_at_20": spread(rows, "ndcg_at_20"),
                              "final_train_loss": spread(rows, "final_train_loss"),
                              "seconds_per_run": round(statistics.mean(r["seconds"] for r in rows), 2),
                              "runs": [dict(r, init_seed=100 + s, mask_seed=5000 + 17 * s, drop_seed=9000 + 31 * s)
                                       for s, r in enumerate(rows)]}
    short = {k: v for k, v in res.items() if k != "loss_per_step"}
    if "seed_spread" in short:
        short["seed_spread"] = {k: v for k, v in short["seed_spread"].items() if k != "runs"}
    print(json.dumps(short))
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
