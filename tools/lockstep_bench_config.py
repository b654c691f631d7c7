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
import statistics  # noqa: F401
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(device, steps, init_seed, mask_seed, drop_seed, dropout, config="c2"):
    return bench.train_probe(device, dropout, steps, (init_seed, mask_seed, drop_seed), config)


def spread(rows, key):
    return dict(bench.spread(rows, key), values=[r[key] for r in rows])


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
        rows = [run(dev, args.steps, *bench.probe_keys(s), args.dropout) for s in range(args.seeds)]
        res["seed_spread"] = {"what": f"{args.seeds} independent runs of the same procedure (init seed 100+s, MLM key 5000+17s, dropout key "
                                      f"9000+31s), {args.steps} steps each",
                              "recall_at_20": spread(rows, "recall_at_20"), "ndcg_at_20": spread(rows, "ndcg_at_20"),
                              "final_train_loss": spread(rows, "final_train_loss"),
                              "seconds_per_run": round(sum(r["seconds"] for r in rows) / len(rows), 2),
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
